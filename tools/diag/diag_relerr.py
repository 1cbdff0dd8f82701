"""Diagnostic: run the fp32 unit tests of tests/test_ops_gpu.py and print the LARGEST relerr each one saw."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.test_ops_gpu as T

seen = []
orig = T.relerr
def rec(a, b):
    v = orig(a, b); seen.append(v); return v
T.relerr = rec

class MP:
    def setenv(self, k, v): os.environ[k] = v

def run(name, fn, *a):
    seen.clear()
    try:
        fn(*a)
        print(f"{name}: relerrs {['%.1e' % v for v in seen]}", flush=True)
    except Exception as e:
        print(f"{name}: FAILED {type(e).__name__} {str(e)[:100]} relerrs {['%.1e' % v for v in seen]}", flush=True)

run("fuse_fwd_bwd_upsample fp32", T.test_fuse_forward_backward_upsample, "fp32")
run("head_fwd_bwd fp32", T.test_head_forward_backward, "fp32")
for case in [(2, 24, 16, 32, 32, 3), (2, 12, 10, 64, 128, 1), (3, 16, 12, 128, 64, 3)]:
    run(f"block_end {case}", T.test_conv_block_end_backward_in_epilogue, case, "fp32")
for case in [(2, 24, 18, 32, 64, 3, 2), (2, 8, 6, 128, 64, 1, 1), (2, 24, 18, 96, 72, 3, 1)]:
    run(f"chain {case}", T.test_conv_bn_relu_chain_forward_backward, case, "fp32", 128, MP())
