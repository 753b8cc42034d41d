import numpy as np, torch, time, os, sys
sys.path.insert(0, os.getcwd())
import audioflux_amd as af
from oracle import ref
from tests import cases
x = np.stack([cases.noise(100+i, 16000*3+77) for i in range(9)])
def run(env_nofused):
    if env_nofused: os.environ['AFX_NO_FUSED']='1'
    else: os.environ.pop('AFX_NO_FUSED', None)
    bft = af.BFT(128, radix2_exp=11, samplate=16000, low_fre=0.0, high_fre=8000.0, slide_length=512,
                 scale_type=af.SpectralFilterBankScaleType.MEL, data_type=af.SpectralDataType.POWER)
    bft.set_result_type(1)
    out = bft.bft_device(torch.from_numpy(x).cuda()); torch.cuda.synchronize()
    return out.cpu().numpy()
fused = run(False); generic = run(True)
rmel, _ = ref.mel_mfcc(x)
def pr(a,b): return np.abs(a-b).max()/np.abs(b).max()
print('fused vs ref', pr(fused, rmel), 'generic vs ref', pr(generic, rmel), 'fused vs generic', pr(fused, generic))
print('nan', np.isnan(fused).sum(), 'shape', fused.shape)
if pr(fused,rmel) > 1e-4:
    d = np.abs(fused-rmel)
    i = np.unravel_index(d.argmax(), d.shape); print('worst at', i, fused[i], rmel[i])
    print('per-row err frame0', (np.abs(fused[0,0]-rmel[0,0])/np.abs(rmel[0,0]).max())[:16])
