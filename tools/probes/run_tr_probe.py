import ctypes, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "tr_probe.so")
lib = ctypes.CDLL(so)
dev = "cuda:0"
def run(addrs, label):
    a = torch.tensor(addrs, dtype=torch.int32, device=dev)
    o = torch.zeros(256, dtype=torch.int16, device=dev)
    rc = lib.tr_probe(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    o = o.cpu().view(64, 4).tolist()
    print(label, "rc", rc)
    for l in range(64):
        print(f"  lane {l:2d} addr {addrs[l]:5d} -> elems {o[l]}")
# A: lane l -> row l of a [64][4] b16 matrix (8 bytes per lane, contiguous)
run([l * 8 for l in range(64)], "A: addr = lane*8")
# B: 16x16 row-major b16 tile (32-byte rows): lane l -> row (l&15)?? try addr = (l&15)*32 + (l>>4)*8
run([(l & 15) * 32 + (l >> 4) * 8 for l in range(64)], "B: addr = (l&15)*32 + (l>>4)*8")
# C: rows of 128 bytes (our K-loop tile): addr = (l&15)*128 + (l>>4)*8
run([(l & 15) * 128 + (l >> 4) * 8 for l in range(64)], "C: addr = (l&15)*128 + (l>>4)*8")
