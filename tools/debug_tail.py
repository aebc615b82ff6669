import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import hmvae_oracle as O
from helpers import offsets, rel_err
from hm_vae_amd import ops
dev = torch.device("cuda:0")
B, T = 64, 8
torch.manual_seed(0)
off = torch.from_numpy(offsets())
tgt = O.synthetic_batch(B, T, seed=5)
def ref(dtype, dec0):
    dec = dec0.to(dtype).requires_grad_(True)
    g6, gR = tgt[0].to(dtype), tgt[1].to(dtype)
    gP = O.fk_positions(gR.reshape(B*T,24,3,3), O.SMPL24_PARENTS, off.to(dtype)).reshape(B,T,72)
    d6 = dec.transpose(1,2).reshape(B*T,24,6)
    R = O.rot6d_to_matrix(d6); P = O.fk_positions(R, O.SMPL24_PARENTS, off.to(dtype))
    l = ((d6.reshape(B,T,-1)-g6)**2).mean() + ((R.reshape(B,T,-1)-gR)**2).mean() + 10*((P.reshape(B,T,-1)-gP)**2).mean()
    l.backward()
    return dec.grad, R.detach(), P.detach(), gP
for scale in (1.0, 0.1):
    dec0 = torch.randn(B,144,T)*scale
    g64, R64, P64, gP64 = ref(torch.float64, dec0)
    g32, R32, P32, gP32 = ref(torch.float32, dec0)
    bp = ops.pad_batch(B)
    def soa(x_bct):
        s = torch.empty(x_bct.shape[1]*x_bct.shape[2], bp, device=dev); ops.aos_to_soa(x_bct.contiguous().to(dev), s, B, 1, x_bct.shape[1]*x_bct.shape[2], bp); return s
    def tsoa(t, C):
        s = torch.empty(C*T, bp, device=dev); ops.aos_to_soa(t.contiguous().to(dev), s, B, T, C, bp); return s
    decs = soa(dec0)
    dd = torch.empty(144*T, bp, device=dev); part = torch.empty((bp//64)*T, 3, device=dev)
    ops.decode_tail_loss(decs, tsoa(tgt[0],144), tsoa(tgt[1],216), tsoa(gP32,72), off.to(dev), T, B, bp, 1/(B*T*144), 1/(B*T*216), 10/(B*T*72), dd, part)
    out = torch.empty(B, 144*T, device=dev); ops.soa_to_aos(dd, out, B, 1, 144*T, bp)
    mine = out.reshape(B,144,T).cpu()
    print("scale", scale, "ddec: hip-vs-f64 %.2e  ref32-vs-f64 %.2e  hip-vs-ref32 %.2e" % (rel_err(mine.numpy(), g64.numpy()), rel_err(g32.numpy(), g64.numpy()), rel_err(mine.numpy(), g32.numpy())))
    c6 = torch.empty(B,T,24,6,device=dev); Ro = torch.empty(B,T,24,3,3,device=dev); Po = torch.empty(B,T,24,3,device=dev)
    ops.decode_tail_fwd(decs, off.to(dev), T, B, bp, c6, Ro, Po)
    print("   R: hip-vs-f64 %.2e ref32-vs-f64 %.2e | P: hip %.2e ref32 %.2e" % (rel_err(Ro.cpu().numpy().reshape(B*T,24,3,3), R64.numpy()), rel_err(R32.numpy(), R64.numpy()), rel_err(Po.cpu().numpy().reshape(B*T,24,3), P64.numpy()), rel_err(P32.numpy(), P64.numpy())))
    # per-element relative to local magnitude percentile
    d = np.abs(mine.numpy()-g64.numpy()); idx = np.unravel_index(d.argmax(), d.shape); print("   worst idx", idx, "hip", mine.numpy()[idx], "f64", g64.numpy()[idx], "f32", g32.numpy()[idx])
