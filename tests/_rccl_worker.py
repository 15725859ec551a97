"""Worker of tests/test_gpu_rccl.py: runs in a process that never imports torch (torch ships its own
librccl; this binding links the ROCm one).  Pure ctypes over the two C ABIs."""
import ctypes as C
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hipl = C.CDLL(os.path.join(ROOT, "c2-ray3dm_amd", "libc2ray_hip.so"), mode=C.RTLD_GLOBAL)
rccl = C.CDLL(os.path.join(ROOT, "c2-ray3dm_amd", "libc2ray_rccl.so"))
sys.path.insert(0, ROOT)


def main():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_capi_only", os.path.join(ROOT, "c2-ray3dm_amd", "_capi.py"))
    capi = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(capi)                      # struct definitions only; no torch import at module level
    p = capi.Params()
    assert hipl.c2r_default_params(C.byref(p)) == 0
    n = 16
    p.mesh[0] = p.mesh[1] = p.mesh[2] = n
    ctx = C.c_void_p()
    assert hipl.c2r_create(C.byref(ctx), C.byref(p)) == 0
    uid = (C.c_char * 128)()
    assert rccl.c2r_rccl_unique_id(uid) == 0
    assert rccl.c2r_rccl_attach(ctx, uid, 0, 1) == 0
    assert rccl.c2r_rccl_attach(ctx, uid, 0, 1) == -2            # already attached
    a = np.random.default_rng(3).random(n ** 3)
    assert hipl.c2r_upload(ctx, 4, a.ctypes.data_as(C.c_void_p)) == 0
    ptr = C.c_void_p()
    assert hipl.c2r_device_ptr(ctx, 4, C.byref(ptr)) == 0
    rccl.c2r_rccl_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    assert rccl.c2r_rccl_allreduce(ctx, ptr, n ** 3, None) == 0   # one rank: the sum is the input
    b = np.empty_like(a)
    assert hipl.c2r_download(ctx, 4, b.ctypes.data_as(C.c_void_p)) == 0
    assert np.array_equal(a, b)
    assert rccl.c2r_rccl_detach(ctx) == 0
    assert rccl.c2r_rccl_detach(ctx) == -2
    hipl.c2r_destroy(ctx)
    print("RCCL_OK")


if __name__ == "__main__":
    main()
