/* C ABI of the data-parallel engine in libvmm_hip.so: gradient exchange over RCCL (xGMI) for the training step of
 * VideoMetamaterials, and the two collectives sharded sampling needs.
 *
 * Reference interface this replaces (vddp.py = /root/reference/denoising_diffusion_pytorch/video_denoising_diffusion_pytorch.py):
 *   - `Accelerator(..., mixed_precision='fp16')` + `accelerator.prepare(model)` + `accelerator.backward(loss)` (main.py:31-34,
 *     vddp.py:1449, 1629): DistributedDataParallel's constructor broadcast and its bucketed gradient all-reduce;
 *   - `broadcast_object_list` of the conditioning matrix and `accelerator.gather` of the padded predictions
 *     (vddp.py:1506-1532, 1745-1749, 1838-1842).
 * SURVEY.md 8(b), last row: init / register_buckets / allreduce_bucket_async / wait_all / finalize.
 *
 * One engine per process (one process per GPU).  The engine owns an RCCL communicator, ONE side stream and a few events; it allocates no
 * device memory and never blocks the host except in vmm_dp_init / vmm_dp_finalize (communicator set-up / tear-down) and
 * vmm_dp_timing (which reads event times: call it after a synchronize).  There is no state outside the handle.  RCCL is bound at run
 * time (dlopen of `rccl_path`, NULL = "librccl.so.1"): a host that already carries an RCCL (PyTorch-ROCm ships one) passes that
 * file's path so that the process holds ONE copy of the library.
 * Return value: 0 on success; > 0 = hipError_t; < 0 = -(1000 + ncclResult_t) for an RCCL failure, -1 bad argument, -2 RCCL could not
 * be loaded.  vmm_dp_last_error(e) names the failing call.
 */
#ifndef VMM_DP_H
#define VMM_DP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vmm_dp_stream_t; /* hipStream_t */
typedef struct vmm_dp_engine vmm_dp_engine;

#define VMM_DP_UNIQUE_ID_BYTES 128 /* sizeof(ncclUniqueId) */

/* Rank 0 creates the rendezvous id and hands its 128 bytes to every rank by whatever channel the host has (a file, a TCP store, MPI). */
int vmm_dp_get_unique_id(const char* rccl_path, void* id_out);

/* Collective over all `world` ranks: joins the communicator on HIP device `device` (made current for the calling thread). */
int vmm_dp_init(vmm_dp_engine** out, const char* rccl_path, int32_t rank, int32_t world, const void* unique_id, int32_t device);

/* The static bucket list of one training plan: bucket i = counts[i] fp32 gradients at ptrs[i] (slices of the flat gradient buffer, reduced
 * IN PLACE, sum).  Replaces any list registered before.  Must be identical (counts) on every rank. */
int vmm_dp_register_buckets(vmm_dp_engine* e, float* const* ptrs, const int64_t* counts, int32_t n);

/* "Everything enqueued on `compute_stream` so far has produced bucket i": the side stream waits for that point, then all-reduces the
 * bucket.  Returns immediately; the backward keeps enqueueing on compute_stream. */
int vmm_dp_allreduce_bucket_async(vmm_dp_engine* e, int32_t i, vmm_dp_stream_t compute_stream);

/* Makes compute_stream wait for every bucket reduction issued since the last wait_all (the optimiser launch goes behind it). */
int vmm_dp_wait_all(vmm_dp_engine* e, vmm_dp_stream_t compute_stream);

/* Measurement: when != 0, bucket reductions are bracketed by timing events and vmm_dp_window_mark(e, 0 / 1, s) stamps the opening /
 * closing of the backward window on the compute stream.  vmm_dp_timing (after a synchronize) returns out[0] = side-stream busy ms of the
 * buckets issued since the window opened, out[1] = the part of it inside the window, out[2] = the window's length in ms. */
int vmm_dp_set_timing(vmm_dp_engine* e, int32_t on);
int vmm_dp_window_mark(vmm_dp_engine* e, int32_t which, vmm_dp_stream_t compute_stream);
int vmm_dp_timing(vmm_dp_engine* e, float* out3);
/* per registered bucket i < n: out[2 i] = ms from the opening window mark to the START of its reduction on the side stream, out[2 i + 1] = to its
 * completion (-1, -1: not reduced since the window opened).  A scaling result below expectation is read from these: late starts = the
 * backward produced the slice late or the side stream was busy, long spans = the collective itself (link bandwidth, contention). */
int vmm_dp_bucket_timing(vmm_dp_engine* e, float* out, int32_t n);

/* Plain collectives ON `stream` (ordered with the caller's work, no side stream):
 *   all-reduce in place; dtype 0 = fp32, 1 = fp64, 2 = int32, 3 = int64; op 0 = sum, 1 = max, 2 = min;
 *   broadcast of `bytes` bytes from `root`, in place;  all-gather of bytes_per_rank bytes from every rank into recv[world][bytes_per_rank]. */
int vmm_dp_allreduce(vmm_dp_engine* e, void* buf, int64_t count, int32_t dtype, int32_t op, vmm_dp_stream_t stream);
int vmm_dp_broadcast(vmm_dp_engine* e, void* buf, int64_t bytes, int32_t root, vmm_dp_stream_t stream);
int vmm_dp_all_gather(vmm_dp_engine* e, const void* send, void* recv, int64_t bytes_per_rank, vmm_dp_stream_t stream);

int vmm_dp_rank(const vmm_dp_engine* e);
int vmm_dp_world(const vmm_dp_engine* e);
/* RCCL's version code (ncclGetVersion) of the library the engine is bound to, 0 if none. */
int vmm_dp_rccl_version(const vmm_dp_engine* e);
/* Text of the last failure on this engine ("" if none); owned by the engine. */
const char* vmm_dp_last_error(const vmm_dp_engine* e);

/* Waits for the side stream, destroys communicator, stream and events, frees the handle. */
int vmm_dp_finalize(vmm_dp_engine* e);

#ifdef __cplusplus
}
#endif
#endif
