import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import audioflux_amd as af
rng = np.random.default_rng(33)
n = 30000
x = (0.1 * rng.standard_normal((4, n))).astype(np.float32)
o = af.CQT(num=84, samplate=44100, low_fre=32.703, bin_per_octave=12, normal_type=af.SpectralFilterBankNormalType.AREA)
xd = torch.from_numpy(x).cuda()
re, im = o.cqt_device(xd)
ch = o.chroma_device(re, im)
torch.cuda.synchronize()
got = np.swapaxes(re.cpu().numpy() + 1j * im.cpu().numpy(), -1, -2)
gch = np.swapaxes(ch.cpu().numpy(), -1, -2)
q = o.cqt(x[0])
print("cqt equal", np.array_equal(got[0], q))
c1 = o.chroma(q)
c2 = o.chroma(q)
print("loop repeat equal", np.array_equal(c1, c2), "batch vs loop maxdiff", np.abs(gch[0] - c1).max())
d = np.argwhere(gch[0] != c1)
print(len(d), d[:10])
p = np.abs(q.astype(np.complex128)) ** 2
fold = np.zeros((12, 84)); 
for j in range(84): fold[j % 12, j] = 1
e = fold @ p; e = e / e.max(0)
print("batch err vs f64", np.abs(gch[0] - e).max(), "loop err", np.abs(c1 - e).max())
ch2 = o.chroma_device(re, im); torch.cuda.synchronize()
print("batch repeat equal", torch.equal(ch, ch2))
ch3 = o.chroma_device(re[0:1].contiguous(), im[0:1].contiguous()); torch.cuda.synchronize()
print("first-clip-only device vs batch equal", torch.equal(ch3[0], ch[0]), "vs loop", np.array_equal(ch3[0].cpu().numpy().T, c1))
print(repr(gch[0][2, 234]), repr(c1[2, 234]), repr(ch3[0].cpu().numpy()[234, 2]))
print("col 234 batch", gch[0][:, 234]); print("col 234 loop ", c1[:, 234])
qq = got[0][:, 234]
print("bins of chroma 2:", [ (abs(qq[j])**2) for j in range(2, 84, 12)])
