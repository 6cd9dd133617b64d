/*
 * td_seam.h — C-ABI of the multi-GPU window sharding (libtd_seam.so): the shard plan and the seam exchange.
 *
 * The reference (xandergos/terrain-diffusion) is single-process and has no counterpart; its scaling mechanism is the algorithm itself: inside a
 * phase every window is independent and between phases a canvas pixel needs the windows that overlap it
 * (terrain_diffusion/training/evaluation/sample_diffusion_base.py:147-168; the multi-phase form terrain_diffusion/inference/world_pipeline.py:1133-1203).
 * SURVEY.md §8e cuts the window grid into a 2-D block mesh, one block per GPU, with one point-to-point exchange per phase and no all-reduce, and
 * §8b proposes `td_halo_exchange(eng, canvas, neighbours[], ncclComm_t, stream)` as its C entry.  This header is that entry.  What crosses a seam
 * is the raw OUTPUT of the windows that reach into a neighbour's region (not accumulator strips): every rank then blends its region in the
 * reference's ascending window order, which keeps the result bit-identical to the single-GPU canvas.
 *
 * Two parts:
 *   * td_seam_plan_*      host arithmetic only (no GPU, no RCCL call): the same partition terrain_diffusion_amd/parallel.py::ShardPlan computes,
 *                         so that a non-Python host can shard through the C-ABI.  tests/test_seam_cpu.py holds the two against each other.
 *   * td_seam_comm_* / td_seam_exchange*   one grouped ncclSend/ncclRecv (RCCL, xGMI) per exchange on a CALLER-SUPPLIED HIP stream — pass the
 *                         engine's stream (td_engine_stream) and window sampling -> exchange -> blend is ordered by the stream, the host
 *                         never waits.  Nothing here synchronises.
 *
 * The library is separate from libtd_engine.so on purpose: the engine carries no RCCL dependency, a one-GPU host never loads librccl.
 * Conventions as in td_engine.h: plain C, opaque handles, 0 on success / negative code on failure with a message in td_seam_last_error().
 * One communicator per GPU / process; not thread-safe per handle.
 */
#ifndef TD_SEAM_H
#define TD_SEAM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { TD_SEAM_OK = 0, TD_SEAM_ERR_ARG = -1, TD_SEAM_ERR_HIP = -2, TD_SEAM_ERR_STATE = -3, TD_SEAM_ERR_RCCL = -5 };
enum { TD_SEAM_ID_BYTES = 128 };   /* = NCCL_UNIQUE_ID_BYTES */

/* window lists a plan can enumerate for a rank (td_seam_plan_windows) */
enum {
    TD_SEAM_OWN = 0,      /* the windows the rank samples, in sampling order (row-major inside its block): the order of `my_tiles` below */
    TD_SEAM_NEEDED = 1,   /* every window that intersects the rank's region, ascending (row, col): the blend order */
    TD_SEAM_SENDS = 2,    /* windows of this rank that another rank's region needs; peer = destination; ascending destination, then blend order */
    TD_SEAM_RECVS = 3     /* windows of other ranks this rank's region needs; peer = source; ascending source, then blend order: the slot order of `recv_tiles` */
};

typedef struct td_seam_plan td_seam_plan;
typedef struct td_seam_comm td_seam_comm;

/* one message of an exchange: `bytes` bytes at `offset` from the base pointer given to td_seam_exchange, to / from rank `peer` */
typedef struct td_seam_msg {
    int32_t peer;
    int32_t reserved;
    int64_t offset;
    int64_t bytes;
} td_seam_msg;

const char* td_seam_last_error(void);

/* ---- the shard plan (host arithmetic; parallel.py::ShardPlan) -------------------------------------------------------------------------
 * H x W canvas, square windows of `tile` pixels every `stride` (0 -> tile/2) with a final window flush with the end
 * (training/evaluation/__init__.py:16-22), `world` ranks on a pr x pc block mesh as square as possible.  extended = 0: a rank's region is the
 * canvas area it OWNS (the regions tile the canvas; single-phase sampler and the last phase).  extended = 1: the bounding box of its own
 * windows — what an intermediate phase of the multi-phase sampler blends, because the next phase cuts every window input out of it. */
int td_seam_plan_create(int H, int W, int tile, int stride, int world, int extended, td_seam_plan** plan);
void td_seam_plan_destroy(td_seam_plan* plan);
/* mesh[0..3] = block rows, block cols, window rows, window cols */
int td_seam_plan_mesh(const td_seam_plan* plan, int32_t mesh[4]);
/* region[0..3] = y0, y1, x0, x1 of the rank's region */
int td_seam_plan_region(const td_seam_plan* plan, int rank, int32_t region[4]);
/* Window origins along one axis (axis 0 = rows, 1 = columns).  Returns the count; fills at most cap entries. */
int td_seam_plan_starts(const td_seam_plan* plan, int axis, int32_t* out, int cap);
/* Enumerates one of the lists above.  Returns the count (>= 0) and fills at most `cap` entries of ij (pairs: window row index, column index)
 * and peer (may be NULL; the rank itself for OWN, the owner for NEEDED). */
int td_seam_plan_windows(const td_seam_plan* plan, int rank, int kind, int32_t* ij, int32_t* peer, int cap);
/* The messages of one exchange for `rank`, with windows of `window_bytes` bytes: send offsets index the rank's OWN-order tile array, receive
 * offsets its RECVS-order slot array.  Windows that are neighbours in the sender's array travel as ONE message; sender and receiver derive
 * the same cuts from the plan.  Returns TD_SEAM_OK and the counts; fills at most cap entries of each list. */
int td_seam_plan_messages(const td_seam_plan* plan, int rank, int64_t window_bytes, td_seam_msg* sends, int* n_sends, td_seam_msg* recvs, int* n_recvs,
                          int cap);

/* ---- the communicator (RCCL) ----------------------------------------------------------------------------------------------------------- */
/* Rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by the host's own means (the Python host: torch.distributed
 * broadcast_object_list; a C host: its launcher / a file / MPI). */
int td_seam_unique_id(void* id128);
/* ncclCommInitRank on `device` (collective over the `world` ranks).  Leaves `device` the calling thread's current device, as hipSetDevice does. */
int td_seam_comm_create(int device, int world, int rank, const void* id128, td_seam_comm** comm);
/* Wraps a communicator the host already has (an ncclComm_t); it is not destroyed with the handle. */
int td_seam_comm_adopt(void* nccl_comm, td_seam_comm** comm);
void td_seam_comm_destroy(td_seam_comm* comm);
/* info[0..2] = world, rank, device */
int td_seam_comm_info(const td_seam_comm* comm, int32_t info[3]);

/* ---- the exchange ---------------------------------------------------------------------------------------------------------------------- */
/* All messages as ONE ncclGroup of ncclSend / ncclRecv on `hip_stream` (device memory; a base may be NULL when its offsets are absolute
 * addresses).  Enqueue-only: returns once the group is launched.  A message to the rank itself is legal (RCCL copies locally).  The group is
 * posted with the communicator's device current; the caller's current device is restored before returning. */
int td_seam_exchange(td_seam_comm* comm, const void* send_base, const td_seam_msg* sends, int n_sends, void* recv_base, const td_seam_msg* recvs,
                     int n_recvs, void* hip_stream);
/* The seam exchange of one phase (parallel.py::exchange_windows): my_tiles = the rank's window outputs in OWN order, recv_tiles = room for
 * the RECVS list, both `window_bytes` per window.  Afterwards window k of RECVS sits in slot k of recv_tiles; together with the rank's own
 * NEEDED windows that is everything its region blends.  The plan's world must equal the communicator's. */
int td_seam_exchange_windows(td_seam_comm* comm, const td_seam_plan* plan, const void* my_tiles, void* recv_tiles, int64_t window_bytes,
                             void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
