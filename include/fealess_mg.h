/*
 * fealess_mg.h -- C ABI of libfealess_mg.so: the multi-GPU host of the template-sharded recognition (BASELINE.json
 * configs[3]: 16000 templates sharded 2000 per GPU, all-gather of the per-GPU top-k detections over xGMI), written in
 * C++ directly on RCCL -- no Python, no torch.distributed -- so that a CadReco C++ caller can reach it.
 *
 * The reference is single-process and has no communication layer; what is preserved across the
 * ranks is Detector::match's ONE global std::sort + std::unique over all templates (linemod/linemod.cpp:1437-1439,
 * the N-template loop :1458) and Recognition()'s refinement of matches[0] (CadReco/obj_reco_lmicp.cpp:111-197).
 *
 * One rank = one process (or thread) = one GPU = one fl_context + fl_detector holding a CONTIGUOUS slice
 * [tid_first, tid_first + tid_count) of the whole bank's template ids together with those templates' depth renders.
 * Every rank is handed the same frames.  Per batch and rank, queued on the context's stream without a host round trip
 * in between:
 *   fl_match_batch_submit -> fl_export_topk_batch (k records per frame, global ids)
 *   -> ncclAllGather of the records (n_ranks * n_frames * k * 20 bytes: latency-bound)
 *   -> fl_select_best_batch (per frame the best of the ranks' first records = matches[0] of the global sort)
 *   -> fl_refine_selected (the owner of the winning template runs the ICP half of Recognition())
 *   -> ncclAllReduce(sum) of the {found, 4x4 pose} rows as INT32 bit patterns (one owner per frame, zeros elsewhere:
 *      the sum is the owner's row bit for bit; a float sum would turn an owner's -0.0 into +0.0)
 * and one stream synchronisation at the end.  A frame whose candidate buffers overflowed on some rank is flagged in the
 * gathered records, so every rank sees it: all ranks grow their buffers (fl_detector_grow_candidates; the outcome is
 * all-reduced, so a rank that cannot grow does not leave the others waiting in the next collective) and the batch runs again.
 *
 * Conventions as in fealess_hip.h: plain pointers and sizes, negative fl_status on failure, fl_mg_last_error() has the text.
 * Every call on a group is COLLECTIVE: all ranks make it, with the same n_frames.
 */
#ifndef FEALESS_MG_H
#define FEALESS_MG_H

#include "fealess_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FL_MG_ID_BYTES 128      /* sizeof(ncclUniqueId) */

typedef struct fl_mg fl_mg;

/* what Recognition() yields for one frame (CadReco/obj_reco_lmicp.cpp:86-204), identical on every rank */
typedef struct {
  int32_t status;               /* FL_OK, or FL_ERR_OVERFLOW: a rank's candidate buffers overflowed and could not be grown */
  int32_t found;                /* 0: vtResult empty; 1: one TObjRecoResult */
  fl_match best;                /* matches[0] of the global sort, GLOBAL template id (template_id = -1: no rank matched) */
  float   pose[16];             /* TObjRecoResult::tWorld2Cam, row-major 4x4 */
} fl_mg_result;

/* ncclGetUniqueId: ONE rank calls it and hands the FL_MG_ID_BYTES bytes to every rank (file, socket, MPI, ...) */
int  fl_mg_unique_id(void *id_out, size_t bytes);
/* ncclCommInitRank on the detector's device (blocks until all n_ranks ranks have called it).  det is finalized and holds
 * the templates [tid_first, tid_first + tid_count) of the whole bank (class-local ids 0 .. tid_count - 1); k = records per
 * frame and rank in the all-gather (Recognition() needs 1; a caller that wants the head of the merged list more). */
int  fl_mg_create(fl_detector *det, const void *id, int n_ranks, int rank, int tid_first, int tid_count, int k, fl_mg **out);
void fl_mg_destroy(fl_mg *mg);
const char *fl_mg_last_error(const fl_mg *mg);
/* CObjRecoLmICP::Recognition for n_frames frames against the whole (sharded) bank.  bgr / depth as for fl_recognize_batch
 * (host or device pointers, the SAME frames on every rank); results (host, n_frames entries) are identical on every rank. */
int  fl_mg_recognize_batch(fl_mg *mg, int n_frames, const uint8_t *const *bgr, const uint16_t *const *depth, int mem,
                           const fl_intrinsics *K, const fl_recognition_params *params, fl_mg_result *results);
/* diagnostics of the last fl_mg_recognize_batch: attempts (1 + buffer growths), bytes all-gathered / all-reduced per attempt */
int  fl_mg_last_stats(const fl_mg *mg, int32_t *attempts, size_t *allgather_bytes, size_t *allreduce_bytes);

#ifdef __cplusplus
}
#endif
#endif /* FEALESS_MG_H */
