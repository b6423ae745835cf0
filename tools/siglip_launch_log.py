import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from bench import load_pkg
pkg = load_pkg()
from llama_cpp_omni_amd import encoders as E
be = pkg.backend(0); be.set_option("graphs", 0)
c = pkg.Context(be)
W = E.siglip2_weights(c, E.SIGLIP2, 2); inp, vit = E.siglip2(c, E.SIGLIP2, W)
c.alloc()
be.graph_compute(c.graph()); be.synchronize()
print("kernels", be.get_stat("kernels_last_graph"))
