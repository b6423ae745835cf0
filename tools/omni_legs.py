import sys, json
sys.path.insert(0, "/root/repo")
import bench
pkg = bench.load_pkg(); be = pkg.backend(0)
print(json.dumps(bench.omni_module_legs(pkg, be)))
