"""The restated ABI (csrc/ggml_abi.h) must agree with the reference headers.

tests/golden/abi.json was produced by oracle/abi_probe.c from the REAL headers under /root/reference/ggml.
Here a twin probe is compiled against OUR header and every shared key is compared.  Also the ctypes mirror.
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

from conftest import GOLDEN, ROOT

TWIN = r'''
#include <stdio.h>
#include <stddef.h>
#include "ggml_abi.h"
#define P(x) printf("  \"%s\": %ld,\n", #x, (long)(x))
int main(void){ printf("{\n");
 P(sizeof(struct ggml_tensor)); P(offsetof(struct ggml_tensor,type)); P(offsetof(struct ggml_tensor,buffer));P(offsetof(struct ggml_tensor,ne));P(offsetof(struct ggml_tensor,nb));
 P(offsetof(struct ggml_tensor,op));P(offsetof(struct ggml_tensor,op_params));P(offsetof(struct ggml_tensor,flags));P(offsetof(struct ggml_tensor,src));P(offsetof(struct ggml_tensor,view_src));P(offsetof(struct ggml_tensor,view_offs));P(offsetof(struct ggml_tensor,data));P(offsetof(struct ggml_tensor,name));P(offsetof(struct ggml_tensor,extra));
 P(sizeof(struct ggml_cgraph));P(offsetof(struct ggml_cgraph,n_nodes));P(offsetof(struct ggml_cgraph,nodes));P(offsetof(struct ggml_cgraph,leafs));P(offsetof(struct ggml_cgraph,use_counts));P(offsetof(struct ggml_cgraph,visited_hash_set));P(offsetof(struct ggml_cgraph,order));
 P(sizeof(struct ggml_backend_i));P(sizeof(struct ggml_backend_buffer_i));P(sizeof(struct ggml_backend_buffer_type_i));P(sizeof(struct ggml_backend_device_i));P(sizeof(struct ggml_backend_reg_i));
 P(sizeof(struct ggml_backend));P(sizeof(struct ggml_backend_buffer));P(sizeof(struct ggml_backend_buffer_type));P(sizeof(struct ggml_backend_device));P(sizeof(struct ggml_backend_reg));P(sizeof(struct ggml_backend_event));P(sizeof(struct ggml_backend_dev_props));P(sizeof(struct ggml_backend_dev_caps));
 P(offsetof(struct ggml_backend,iface));P(offsetof(struct ggml_backend,device));P(offsetof(struct ggml_backend,context));
 P(offsetof(struct ggml_backend_buffer,buft));P(offsetof(struct ggml_backend_buffer,context));P(offsetof(struct ggml_backend_buffer,size));P(offsetof(struct ggml_backend_buffer,usage));
 P(offsetof(struct ggml_backend_reg,iface));P(offsetof(struct ggml_backend_reg,context));
 P(offsetof(struct ggml_backend_dev_props,memory_free));P(offsetof(struct ggml_backend_dev_props,type));P(offsetof(struct ggml_backend_dev_props,device_id));P(offsetof(struct ggml_backend_dev_props,caps));
 P(GGML_BACKEND_API_VERSION);
 P(GGML_OP_NONE);P(GGML_OP_DUP);P(GGML_OP_ADD);P(GGML_OP_ADD1);P(GGML_OP_SUB);P(GGML_OP_MUL);P(GGML_OP_DIV);P(GGML_OP_SQR);P(GGML_OP_SQRT);P(GGML_OP_SUM_ROWS);P(GGML_OP_ARGMAX);P(GGML_OP_REPEAT);P(GGML_OP_CONCAT);P(GGML_OP_NORM);P(GGML_OP_RMS_NORM);P(GGML_OP_GROUP_NORM);P(GGML_OP_L2_NORM);P(GGML_OP_MUL_MAT);P(GGML_OP_MUL_MAT_ID);P(GGML_OP_SCALE);P(GGML_OP_SET);P(GGML_OP_CPY);P(GGML_OP_CONT);P(GGML_OP_RESHAPE);P(GGML_OP_VIEW);P(GGML_OP_PERMUTE);P(GGML_OP_TRANSPOSE);P(GGML_OP_GET_ROWS);P(GGML_OP_SET_ROWS);P(GGML_OP_DIAG_MASK_INF);P(GGML_OP_SOFT_MAX);P(GGML_OP_ROPE);P(GGML_OP_CLAMP);P(GGML_OP_IM2COL);P(GGML_OP_CONV_2D);P(GGML_OP_POOL_1D);P(GGML_OP_POOL_2D);P(GGML_OP_UPSCALE);P(GGML_OP_PAD);P(GGML_OP_ARANGE);P(GGML_OP_TIMESTEP_EMBEDDING);P(GGML_OP_ARGSORT);P(GGML_OP_LEAKY_RELU);P(GGML_OP_FLASH_ATTN_EXT);P(GGML_OP_UNARY);P(GGML_OP_GLU);P(GGML_OP_COUNT);
 P(GGML_UNARY_OP_ABS);P(GGML_UNARY_OP_NEG);P(GGML_UNARY_OP_TANH);P(GGML_UNARY_OP_ELU);P(GGML_UNARY_OP_RELU);P(GGML_UNARY_OP_SIGMOID);P(GGML_UNARY_OP_GELU);P(GGML_UNARY_OP_GELU_QUICK);P(GGML_UNARY_OP_SILU);P(GGML_UNARY_OP_EXP);P(GGML_UNARY_OP_GELU_ERF);P(GGML_UNARY_OP_COUNT);
 P(GGML_GLU_OP_REGLU);P(GGML_GLU_OP_GEGLU);P(GGML_GLU_OP_SWIGLU);P(GGML_GLU_OP_SWIGLU_OAI);P(GGML_GLU_OP_GEGLU_ERF);P(GGML_GLU_OP_GEGLU_QUICK);
 P(GGML_TYPE_F32);P(GGML_TYPE_F16);P(GGML_TYPE_Q4_0);P(GGML_TYPE_Q8_0);P(GGML_TYPE_Q8_1);P(GGML_TYPE_Q4_K);P(GGML_TYPE_Q5_K);P(GGML_TYPE_Q6_K);P(GGML_TYPE_Q8_K);P(GGML_TYPE_I8);P(GGML_TYPE_I16);P(GGML_TYPE_I32);P(GGML_TYPE_I64);P(GGML_TYPE_F64);P(GGML_TYPE_BF16);P(GGML_TYPE_COUNT);
 P(GGML_PREC_F32);P(GGML_ROPE_TYPE_NEOX);P(GGML_ROPE_TYPE_MROPE);P(GGML_ROPE_TYPE_VISION);P(GGML_KQ_MASK_PAD);
 P(GGML_BACKEND_DEVICE_TYPE_CPU);P(GGML_BACKEND_DEVICE_TYPE_GPU);P(GGML_BACKEND_BUFFER_USAGE_WEIGHTS);P(GGML_BACKEND_BUFFER_USAGE_COMPUTE);
 P(GGML_STATUS_ALLOC_FAILED);P(GGML_STATUS_FAILED);P(GGML_STATUS_SUCCESS);P(GGML_STATUS_ABORTED);
 P(GGML_TENSOR_FLAG_INPUT);P(GGML_TENSOR_FLAG_OUTPUT);P(GGML_LOG_LEVEL_INFO);P(GGML_LOG_LEVEL_WARN);P(GGML_LOG_LEVEL_ERROR);P(GGML_LOG_LEVEL_DEBUG);
 printf("  \"_end\": 0\n}\n"); return 0; }
'''


def test_restated_abi_matches_reference_headers():
    gold = json.load(open(os.path.join(GOLDEN, "abi.json")))
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "twin.c")
        open(src, "w").write(TWIN)
        exe = os.path.join(td, "twin")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "llama.cpp-omni_amd", "csrc"), src, "-o", exe])
        mine = json.loads(subprocess.check_output([exe]))
    assert set(mine) == set(gold), set(mine) ^ set(gold)
    bad = {k: (mine[k], gold[k]) for k in gold if mine[k] != gold[k]}
    assert not bad, bad


def test_ctypes_mirror_matches_abi(pkg):
    gold = json.load(open(os.path.join(GOLDEN, "abi.json")))
    g = sys.modules["llama_cpp_omni_amd.ggml"]
    assert C.sizeof(g.ggml_tensor) == gold["sizeof(struct ggml_tensor)"]
    assert g.ggml_tensor.data.offset == gold["offsetof(struct ggml_tensor,data)"]
    assert g.ggml_tensor.src.offset == gold["offsetof(struct ggml_tensor,src)"]
    assert g.ggml_tensor.op_params.offset == gold["offsetof(struct ggml_tensor,op_params)"]
    assert C.sizeof(g.ggml_cgraph) == gold["sizeof(struct ggml_cgraph)"]
    assert g.ggml_cgraph.nodes.offset == gold["offsetof(struct ggml_cgraph,nodes)"]
    assert C.sizeof(g.backend_i) == gold["sizeof(struct ggml_backend_i)"]
    assert C.sizeof(g.buffer_i) == gold["sizeof(struct ggml_backend_buffer_i)"]
    assert C.sizeof(g.buft_i) == gold["sizeof(struct ggml_backend_buffer_type_i)"]
    assert C.sizeof(g.device_i) == gold["sizeof(struct ggml_backend_device_i)"]
    assert C.sizeof(g.reg_i) == gold["sizeof(struct ggml_backend_reg_i)"]
    assert C.sizeof(g.backend_t) == gold["sizeof(struct ggml_backend)"]
    assert C.sizeof(g.buffer_t) == gold["sizeof(struct ggml_backend_buffer)"]
    assert C.sizeof(g.reg_t) == gold["sizeof(struct ggml_backend_reg)"]
    assert C.sizeof(g.dev_props) == gold["sizeof(struct ggml_backend_dev_props)"]
    for name, val in (("MUL_MAT", "GGML_OP_MUL_MAT"), ("ROPE", "GGML_OP_ROPE"), ("FLASH_ATTN_EXT", "GGML_OP_FLASH_ATTN_EXT"), ("GLU", "GGML_OP_GLU"),
                      ("SET_ROWS", "GGML_OP_SET_ROWS"), ("GET_ROWS", "GGML_OP_GET_ROWS"), ("SOFT_MAX", "GGML_OP_SOFT_MAX"), ("RMS_NORM", "GGML_OP_RMS_NORM"),
                      ("CPY", "GGML_OP_CPY"), ("CONT", "GGML_OP_CONT"), ("UNARY", "GGML_OP_UNARY"), ("SCALE", "GGML_OP_SCALE")):
        assert getattr(g.OP, name) == gold[val]
