import ctypes as C, numpy as np
from oracle import restate
lib=C.CDLL('audioflux_amd/lib/libaudioflux_mi355x.so')
class Band(C.Structure):
    _fields_=[('num',C.c_int),('tapsA',C.c_int),('tapsB',C.c_int),('startA',C.c_int*64),('startB',C.c_int*64),('rowA',C.c_int*64),('rowB',C.c_int*64),('wA',C.POINTER(C.c_float)),('wB',C.POINTER(C.c_float)),('split',C.c_int),('segIdx',C.c_uint*128)]
for (num,N,sr) in [(128,2048,16000),(128,2048,32000),(80,2048,16000),(64,2048,16000),(40,2048,16000),(128,2048,44100)]:
    bank,_,_=restate.mel_bank(num,N,sr,0,sr/2)
    F=N//2+1
    b=Band()
    rc=lib.afx_bandplan_build(bank.ctypes.data_as(C.POINTER(C.c_float)),num,F,C.byref(b))
    sA=np.array(b.startA); sB=np.array(b.startB); rA=np.array(b.rowA); rB=np.array(b.rowB)
    okA=all(len(set((sA[h*32:(h+1)*32]//2)%32))==32 for h in range(2)) and (sA%2==0).all(); okB=all(len(set((sB[h*32:(h+1)*32]//2)%32))==32 for h in range(2)) and (sB%2==0).all()
    # reconstruct bank from plan
    wA=np.ctypeslib.as_array(b.wA,(b.tapsA,64)); wB=np.ctypeslib.as_array(b.wB,(b.tapsB,64))
    rec=np.zeros_like(bank)
    for l in range(64):
        if rA[l]>=0:
            for t in range(b.tapsA):
                if sA[l]+t<F: rec[rA[l],sA[l]+t]+=wA[t,l]
        if rB[l]>=0:
            for t in range(b.tapsB):
                if sB[l]+t<F: rec[rB[l],sB[l]+t]+=wB[t,l]
    print(num,sr,'rc',rc,'tapsA',b.tapsA,'tapsB',b.tapsB,'conflict-free',okA,okB,'bank reconstructed',np.array_equal(rec,bank), 'rows covered', sorted(set(rA[rA>=0])|set(rB[rB>=0]))==list(range(num)))
