"""Stand-alone timing of the recomputing attention-block backward kernels on the Lagrangian shapes (batch 4, 11 frames):

    python tools/bench_attn_bwd.py [temporal|linear] [HW] [ntok] [reps]

(A/B of two builds on one box: VMM_LIB_PATH=$PWD/videometamaterials_amd/libvmm_hip_ab.so, see tools/build_ab.py.)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from videometamaterials_amd import _native as N  # noqa: E402
from videometamaterials_amd import hostmath  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "temporal"
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 96 * 96
ntok = int(sys.argv[3]) if len(sys.argv) > 3 else 11
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
B, T, Cc, heads, hid = 4, 11, 64, 8, 256
dev = torch.device("cuda", 0)
lib = N.lib()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pack(w2d, transposed=False, fmt=2):
    co, ci = w2d.shape
    wg = w2d.contiguous().to(dev)
    packed = torch.zeros((co + 31) // 32 * 32 * ((ci + 31) // 32 * 32), device=dev)
    job = (N.PackJob * 1)()
    j = job[0]
    j.torch_w, j.packed = wg.data_ptr(), packed.data_ptr()
    if transposed:
        j.TH, j.TW, j.C, j.Cp, j.N, j.sn, j.sc = 1, 1, co, co, ci, 1, ci
    else:
        j.TH, j.TW, j.C, j.Cp, j.N, j.sn, j.sc = 1, 1, ci, ci, co, ci, 1
    j.fmt = fmt
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(dev)
    N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, packed.numel(), 0, s), "pack")
    return packed


g = torch.Generator().manual_seed(1)
rows = B * T * HW
x = torch.randn(rows, Cc, generator=g).to(dev)
dout = torch.randn(rows, Cc, generator=g).to(dev)
gamma = (1 + 0.2 * torch.randn(Cc, generator=g)).to(dev)
wqkv, wout = torch.randn(3 * hid, Cc, generator=g) / 8, torch.randn(Cc, hid, generator=g) / 16
wq, woT, wo3 = pack(wqkv), pack(wout, transposed=True), pack(wout, fmt=3)
bias = torch.randn(heads, T, T, generator=g).to(dev)
rot = hostmath.rotary_table(T, 32).to(dev)
ek, ev = torch.randn(B, max(ntok, 1), hid, generator=g).to(dev), torch.randn(B, max(ntok, 1), hid, generator=g).to(dev)
dqkv = torch.empty(rows, 3 * hid, device=dev)
stats = torch.empty(rows, 2, device=dev)
dwo, dbo = torch.zeros(hid, Cc, device=dev), torch.zeros(Cc, device=dev)
dbias = torch.zeros(heads, T, T, device=dev)
dek, dev_ = torch.zeros_like(ek), torch.zeros_like(ev)
d = N.AttnBlockBwd()
d.x, d.ldx, d.gamma, d.wqkv_frag, d.wout_t_frag = x.data_ptr(), Cc, gamma.data_ptr(), wq.data_ptr(), woT.data_ptr()
if ntok:
    d.ek, d.ev, d.ntok = ek.data_ptr(), ev.data_ptr(), ntok
d.bias, d.bias_on_cond, d.rot_tab = bias.data_ptr(), 1 if ntok == T else 0, rot.data_ptr()
d.dout, d.lddo, d.dqkv, d.lddqkv, d.ln_stats = dout.data_ptr(), Cc, dqkv.data_ptr(), 3 * hid, stats.data_ptr()
d.dwout_packed, d.dbout, d.dbias, d.dek, d.dev = dwo.data_ptr(), dbo.data_ptr(), dbias.data_ptr(), dek.data_ptr(), dev_.data_ptr()
d.B, d.T, d.HW, d.C, d.heads, d.q_scale, d.eps = B, T, HW, Cc, heads, 32 ** -0.5, 1e-5
if kind == "temporal":
    ws = torch.empty(lib.vmm_temporal_block_bwd_workspace(B, T, HW, Cc, heads, ntok), device=dev)
    fn = lib.vmm_temporal_block_bwd_bf16x3
else:
    fws = torch.empty(lib.vmm_linattn_block_workspace(B, T, HW), device=dev)
    out = torch.empty_like(x)
    N.check(lib.vmm_linattn_block_bf16x3(x.data_ptr(), Cc, gamma.data_ptr(), wq.data_ptr(), wo3.data_ptr(), dbo.data_ptr(), d.ek, d.ev, ntok, fws.data_ptr(),
                                         out.data_ptr(), Cc, B, T, HW, Cc, heads, C.c_float(1e-5), s), "forward")
    d.fwd_workspace = fws.data_ptr()
    ws = torch.empty(lib.vmm_linattn_block_bwd_workspace(B, T, HW, Cc, heads, ntok), device=dev)
    fn = lib.vmm_linattn_block_bwd_bf16x3
d.workspace = ws.data_ptr()
for _ in range(3):
    N.check(fn(C.byref(d), s), kind)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn(C.byref(d), s)
e1.record()
torch.cuda.synchronize()
print(f"{kind} block backward, B={B} T={T} HW={HW} ntok={ntok}: {e0.elapsed_time(e1) / reps:.3f} ms per call (kernel + partial sums)  lib={os.environ.get('VMM_LIB_PATH', 'default')}")
