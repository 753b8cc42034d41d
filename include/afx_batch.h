/* afx_batch.h -- ADDITIVE entry points of libaudioflux_mi355x.so (not part of
 * the reference API).  The reference interface is one clip per call through
 * host pointers, which on a GPU is bounded by PCIe and launch latency; these
 * calls expose the same transforms batched over clips and/or on buffers that
 * already live in HBM.  Results are identical to looping the legacy call (exception: the batched
 * cepstrogram at n_fft 2048 / 4096 runs its own kernels and agrees to the parity bar, 1e-5).
 *
 * Device pointers are plain `float*` HBM addresses (e.g. torch.Tensor.data_ptr())
 * and `hipStream` is a hipStream_t passed as void*, used as given (NULL is HIP's
 * default stream, which is what torch.cuda.current_stream().cuda_stream is
 * unless the caller switched streams).  Device variants are asynchronous on that stream; host variants
 * return with the result in host memory.  All return 0 or a negative status
 * (see afx_last_error()).
 */
#ifndef AFX_BATCH_H
#define AFX_BATCH_H

#include "bft_algorithm.h"
#include "cepstrogram_algorithm.h"
#include "cqt_algorithm.h"
#include "cwt_algorithm.h"
#include "feature/xxcc_algorithm.h"
#include "pwt_algorithm.h"
#include "reassign_algorithm.h"
#include "spectrogram_algorithm.h"
#include "stft_algorithm.h"
#include "synsq_algorithm.h"
#include "wsst_algorithm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 0 when a gfx950 device is usable; negative otherwise (no CPU fallback exists) */
int afx_runtime_status(void);
const char *afx_last_error(void);
/* Number of failures reported on the CALLING THREAD so far.  The reference's compute entry points
 * are `void` (bftObj_bft, stftObj_stft, cwtObj_cwt, ...): a HIP failure inside one (out of memory,
 * launch error) cannot be returned, so it is recorded -- message in afx_last_error(), one line on
 * stderr -- and this counter advances.  A caller that samples it before and after a void call knows
 * whether the (pre-zeroed) outputs are results; the audioflux_amd wrappers raise RuntimeError. */
int afx_error_count(void);
int afx_device_count(void);
/* Device ordinal for objects created FROM NOW ON, by any thread (HIP's current device is per
 * thread: every constructor makes this device current first).  An existing object stays on the
 * device it was built on: each compute entry point makes the object's device current for the
 * calling thread (its stream's device), so objects on different devices can be mixed in one
 * process and used from any thread -- one object by one thread at a time, like the reference. */
int afx_set_device(int ordinal);
/* "audioflux_mi355x <version> gfx950" */
const char *afx_version(void);

/* Measurement aid (bench.py: roofline.clock_mhz_this_run): launches ONE sleeping wave on `hipStream` that keeps writing
 * dOut2[0] = shader-clock cycles and dOut2[1] = reference-clock ticks (rate returned in *wallClockKHz) since its start,
 * until the device word *dStop becomes non-zero (write it from another stream) or maxSeconds (<= 60) have passed.
 * dOut2[0] / dOut2[1] * wallClockKHz = the shader clock in kHz the device held while the kernels launched in between
 * ran on its other compute units.  No reference counterpart. */
int afx_clock_probe_start(void *hipStream, unsigned long long *dOut2, const unsigned *dStop, double maxSeconds, int *wallClockKHz);

/* which execution plan a BFT object (bftObj_new, or the one inside a spectrogram object) got:
 * 0 = size-generic kernels, 1 = fused STFT -> filter-bank kernel (n_fft 2048) with whole rows per
 * lane slot, 2 = the same kernel with rows cut into segments (banks whose rows exceed the compiled
 * tap variants), 101 / 201 = the fused n_fft 1024 / 4096 kernels.  Diagnostic only. */
int bftObj_fusedPlanKind(BFTObj bftObj);

/* batch clips of dataLength samples, host pointers:
 * dataArr[batch*dataLength] -> mRealArr3[batch*T*num] (+ mImageArr3 when the
 * result type is complex).  Same as calling bftObj_bft per clip. */
int bftObj_bftBatch(BFTObj bftObj, const float *dataArr, int batch, int dataLength,
                    float *mRealArr3, float *mImageArr3);

/* the same on HBM-resident buffers; clip b starts at dData + b*clipStride */
int bftObj_bftBatchDevice(BFTObj bftObj, const float *dData, int batch, int dataLength,
                          long long clipStride, float *dReal, float *dImag, void *hipStream);

/* cepstral coefficients of rows frames on HBM-resident buffers:
 * dIn[rows*num] -> dOut[rows*ccNum] */
int xxccObj_xxccDevice(XXCCObj xxccObj, const float *dIn, long long rows, int ccNum,
                       CepstralRectifyType *rectifyType, float *dOut, void *hipStream);

/* the north-star path in one call: batched STFT -> filter bank -> cepstra.
 * dMel (batch*T*num) may be NULL when only the cepstra are wanted;
 * dCc is batch*T*ccNum.  bft must be in real result mode (type 1). */
int afx_bftXxccBatchDevice(BFTObj bftObj, XXCCObj xxccObj, const float *dData, int batch,
                           int dataLength, long long clipStride, int ccNum,
                           CepstralRectifyType *rectifyType, float *dMel, float *dCc,
                           void *hipStream);
/* how many afx_bftXxccBatchDevice calls of the calling thread ran as ONE kernel launch so far (the others took the bank
 * kernel + the cepstrum kernel).  Diagnostic only. */
long long afx_bftXxccOneLaunchCount(void);

/* xxccObj_xxcc (feature/xxcc_algorithm.h) with the row count passed explicitly instead of
 * through xxccObj_setTimeLength: mDataArr1[rows*num] -> mDataArr2[rows*ccNum], host pointers */
int xxccObj_xxccBatch(XXCCObj xxccObj, const float *mDataArr1, long long rows, int ccNum,
                      CepstralRectifyType *rectifyType, float *mDataArr2);

/* ---- CWT (BASELINE config 4) --------------------------------------------------------
 * chunks of 2^radix2Exp samples, chunk c at data + c*chunkStride ->
 * real/imag [chunks][num][2^radix2Exp] (row 0 = highest frequency, as cwtObj_cwt).
 * Same as calling cwtObj_cwt (cwt_algorithm.h) per chunk. */
int cwtObj_cwtBatch(CWTObj cwtObj, const float *dataArr, int chunks, float *mRealArr3,
                    float *mImageArr3);
int cwtObj_cwtBatchDevice(CWTObj cwtObj, const float *dData, int chunks, long long chunkStride,
                          float *dReal, float *dImag, void *hipStream);
/* the d/dt variant (cwtObj_cwtDet); cwtObj_enableDet(obj, 1) must have been called */
int cwtObj_cwtDetBatchDevice(CWTObj cwtObj, const float *dData, int chunks, long long chunkStride,
                             float *dReal, float *dImag, void *hipStream);

/* ---- CQT + chroma (BASELINE config 5) -----------------------------------------------
 * batch clips of dataLength samples -> real/imag [batch][T, num], T = cqtObj_calTimeLength.
 * Same as calling cqtObj_cqt (cqt_algorithm.h) per clip; all clips of the batch go through
 * each octave of the recursion in one launch.  Objects created with isContinue = 1 carry one signal's
 * tail from call to call: the batch calls return AFX_ERR_UNSUPPORTED (-4) for them. */
int cqtObj_cqtBatch(CQTObj cqtObj, const float *dataArr, int batch, int dataLength,
                    float *mRealArr, float *mImageArr);
int cqtObj_cqtBatchDevice(CQTObj cqtObj, const float *dData, int batch, int dataLength,
                          long long clipStride, float *dReal, float *dImag, void *hipStream);
/* cqtObj_chroma on `rows` (= batch*T) HBM-resident CQT frames -> dData[rows*chromaNum];
 * the optional parameters keep the meaning and defaults of cqtObj_chroma */
int cqtObj_chromaBatchDevice(CQTObj cqtObj, int *chromaNum, SpectralDataType *dataType,
                             ChromaDataNormalType *normType, const float *dReal,
                             const float *dImag, long long rows, float *dData, void *hipStream);

/* cqtObj_cqtBatchDevice followed by cqtObj_chromaBatchDevice on its output, as one call: the clips go through
 * the octave ladder in passes (AFX_CQT_CHUNK clips; default: the fewest equal passes of <= 448 MB of output) and each pass's chroma is taken
 * while its CQT rows are still cached.  dChroma [batch][T, chromaNum]; results identical to the two calls. */
int cqtObj_cqtChromaBatchDevice(CQTObj cqtObj, const float *dData, int batch, int dataLength, long long clipStride,
                                float *dReal, float *dImag, int *chromaNum, SpectralDataType *dataType,
                                ChromaDataNormalType *normType, float *dChroma, void *hipStream);

/* ---- cepstrogram ----------------------------------------------------------------------
 * batch clips -> dOut1/dOut2/dOut3 [batch][T, N/2+1] (cepstrum, envelope, details; any may
 * be NULL).  Same as calling cepstrogramObj_cepstrogram (cepstrogram_algorithm.h) per clip. */
int cepstrogramObj_cepstrogramBatchDevice(CepstrogramObj cepstrogramObj, int cepNum,
                                          const float *dData, int batch, int dataLength,
                                          long long clipStride, float *dOut1, float *dOut2,
                                          float *dOut3, void *hipStream);

/* ---- PWT: as cwtObj_cwtBatchDevice, bands in ascending order --------------------------------- */
int pwtObj_pwtBatchDevice(PWTObj pwtObj, const float *dData, int chunks, long long chunkStride,
                          float *dReal, float *dImag, void *hipStream);

/* ---- WSST: chunks of 2^radix2Exp samples -> squeezed coefficients ADDED to dReal1/dImag1
 * [chunks][num][2^radix2Exp] (zero them for a plain transform); dReal2/dImag2 (both or neither
 * NULL) receive the CWT itself.  Same as calling wsstObj_wsst (wsst_algorithm.h) per chunk. */
int wsstObj_wsstBatchDevice(WSSTObj wsstObj, const float *dData, int chunks, long long chunkStride,
                            float *dReal1, float *dImag1, float *dReal2, float *dImag2, void *hipStream);

/* ---- reassignment: batch clips -> reassigned coefficients ADDED to dReal1/dImag1 [batch][T, F]
 * (zero them first; dImag1 may be NULL in amplitude result mode); dReal2/dImag2 (both or neither
 * NULL) receive the plain STFT.  Same as calling reassignObj_reassign per clip. */
int reassignObj_reassignBatchDevice(ReassignObj reassignObj, const float *dData, int batch, int dataLength,
                                    long long clipStride, float *dReal1, float *dImag1, float *dReal2,
                                    float *dImag2, void *hipStream);

/* ---- spectrogram object ------------------------------------------------------------------
 * batch clips of dataLength samples -> dSpect [batch][T, num] (T = frames of ONE clip without the
 * streaming tail).  Same as calling spectrogramObj_spectrogram (spectrogram_algorithm.h) per
 * clip on a non-continuing object; mel / bark / erb run the fused STFT -> filter-bank kernels. */
int spectrogramObj_spectrogramBatchDevice(SpectrogramObj spectrogramObj, const float *dData, int batch,
                                          int dataLength, long long clipStride, float *dSpect,
                                          void *hipStream);

/* ---- STFT / inverse STFT ---------------------------------------------------------------
 * batch clips of dataLength samples -> dReal/dImag [batch][T, fftLength] (all fftLength bins),
 * T = stftObj_calTimeLength of ONE clip.  The padding switches of the object apply per clip;
 * the streaming tail (isContinue) is not used: clips of a batch are independent signals.
 * Same as calling stftObj_stft (stft_algorithm.h) per clip. */
int stftObj_stftBatchDevice(STFTObj stftObj, const float *dData, int batch, int dataLength,
                            long long clipStride, float *dReal, float *dImag, void *hipStream);
/* dReal/dImag [batch][nLength, fftLength] -> dData[b*dataStride + j], j < (nLength-1)*slideLength
 * + fftLength.  dData is read-modify-write (zero it for a plain inverse), like stftObj_istft. */
int stftObj_istftBatchDevice(STFTObj stftObj, const float *dReal, const float *dImag, int batch,
                             int nLength, int type, float *dData, long long dataStride,
                             void *hipStream);

/* On-wire format of (gathered) feature tensors: NumPy .npy v1.0, little-endian float32, C order
 * (readable with numpy.load / mmap by any consumer).  Host pointer; returns 0, -1 on an I/O error,
 * -6 on bad arguments. */
int afx_write_npy_f32(const char *path, const float *data, int ndim, const long long *shape);

/* ---- the exchange step (multi-GPU): feature slabs -> root over RCCL / xGMI -------------------
 * One process per GPU, clips sharded contiguously (rank r owns clips [r*ceil(B/G), ...)): the
 * transforms need no collective, the gathered tensor is the ranks' slabs back to back.  RCCL is
 * bound at run time (dlopen of librccl.so.1; a copy already mapped by the process is shared), so
 * single-GPU deployments never load it.  Bootstrap: rank 0 calls afx_comm_get_unique_id and
 * hands the 128 bytes to the other ranks by whatever the launcher offers (MPI, a file, the
 * torch.distributed store: audioflux_amd/dist.py), then every rank calls afx_comm_create -- the
 * communicator is bound to the calling thread's CURRENT HIP device (one process per GPU: select the
 * rank's GPU with afx_set_device / hipSetDevice first); afx_gather returns -6 when `hipStream` lives
 * on another device, and leaves the caller's current device as it found it.
 * afx_gather is asynchronous on `hipStream` like ncclGather (rccl.h:745): `count` floats from
 * dSend on every rank, worldSize*count floats into dRecv on `root` (dRecv is ignored elsewhere).
 * All return 0 or a negative status (-4: RCCL not installed). */
#define AFX_COMM_ID_BYTES 128
typedef struct AfxComm *AfxCommObj;
int afx_comm_get_unique_id(void *id /* out: AFX_COMM_ID_BYTES */);
int afx_comm_create(AfxCommObj *comm, int worldSize, int rank, const void *id);
int afx_comm_world_size(AfxCommObj comm);
int afx_comm_rank(AfxCommObj comm);
int afx_gather(AfxCommObj comm, const float *dSend, long long count, float *dRecv, int root, void *hipStream);
void afx_comm_free(AfxCommObj comm);

#ifdef __cplusplus
}
#endif
#endif /* AFX_BATCH_H */
