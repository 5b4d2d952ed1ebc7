"""which XCDs the workgroups of a CU-masked stream land on: contiguous ranges of 32 mask bits, and bits i with i % 8 == r"""
import ctypes, sys, os, torch
sys.path.insert(0, os.getcwd())
from amcontrast3d_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
torch.cuda.init(); torch.zeros(1, device=dev)
ncu = torch.cuda.get_device_properties(dev).multi_processor_count
print("CUs", ncu)
def probe(bits, tag):
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint * words)()
    for b in bits:
        mask[b >> 5] |= 1 << (b & 31)
    h = ctypes.c_void_p()
    _lib.check(lib.amc3d_stream_create_cu_mask(ctypes.byref(h), mask, words), "mask")
    out = torch.full((2048,), -1, dtype=torch.int32, device=dev)
    _lib.check(lib.amc3d_probe_xcc_ids(2048, ctypes.c_void_p(out.data_ptr()), h), "probe")
    torch.cuda.synchronize()
    hist = torch.bincount(out.cpu().long(), minlength=8).tolist()
    print(f"{tag}: workgroups per XCD {hist}")
    lib.amc3d_stream_destroy(h)
probe(range(ncu), "all CUs")
for r in range(0, ncu, 32):
    probe(range(r, min(r + 32, ncu)), f"bits {r}..{r + 31}")
for r in range(8):
    probe(range(r, ncu, 8), f"bits = {r} mod 8")
probe(range(0, 8), "bits 0..7")
probe(range(0, 16), "bits 0..15")
