// td_seam.cpp — libtd_seam.so: the C-ABI of include/td_seam.h.  Host code only (no kernels): the shard plan is integer arithmetic, the
// exchange is one grouped ncclSend/ncclRecv on the caller's stream.  Python twin of the plan: terrain_diffusion_amd/parallel.py::ShardPlan
// (tests/test_seam_cpu.py compares them list by list); Python twin of the exchange: parallel.py::exchange_windows.
// Reference: none (single process).  The loops this shards: training/evaluation/sample_diffusion_base.py:147-168,
// inference/world_pipeline.py:1133-1203; window origins: training/evaluation/__init__.py:16-22.
#include "../../include/td_seam.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

using Win = std::pair<int, int>;   // (window row index, window column index)

// window origins along one axis: 0, stride, 2*stride, ... while the window fits, plus one window flush with the end (geometry.py::tile_starts)
std::vector<int> starts_of(int length, int tile, int stride) {
    const int last = length > tile ? length - tile : 0;
    std::vector<int> s;
    for (int k = 0; k * stride <= last; ++k) s.push_back(k * stride);
    if (s.back() < last) s.push_back(last);
    return s;
}

}  // namespace

struct td_seam_plan {
    int H, W, tile, stride, world, extended;
    int pr, pc;
    std::vector<int> hs, ws, row_cuts, col_cuts;
    std::vector<int> owner;                  // [window row][window col] -> rank
    std::vector<std::vector<Win>> own;       // per rank, sampling order
    std::vector<std::vector<Win>> needed;    // per rank, blend order
    std::vector<std::array<int, 4>> regions; // per rank: y0, y1, x0, x1
    std::vector<std::vector<int>> local;     // per rank: [window row][window col] -> index in own[rank], -1 elsewhere

    int nr() const { return (int)hs.size(); }
    int nc() const { return (int)ws.size(); }
    int owner_of(const Win& w) const { return owner[w.first * nc() + w.second]; }
    // windows of rank s that rank d's region needs, in d's blend order (parallel.py: plan.sends[(s, d)])
    std::vector<Win> crossing(int s, int d) const {
        std::vector<Win> out;
        for (const Win& w : needed[d])
            if (owner_of(w) == s) out.push_back(w);
        return out;
    }
};

struct td_seam_comm {
    ncclComm_t comm;
    int world, rank, device;
    bool owned;
};

namespace {

// pr x pc = world, blocks as square as possible, never more parts than windows along an axis; the first best split wins (parallel.py::mesh_shape,
// same double arithmetic so that ties fall the same way)
bool mesh_shape(int world, int nr, int nc, int* pr_out, int* pc_out) {
    bool have = false;
    double best = 0.0;
    for (int pr = 1; pr <= world; ++pr) {
        if (world % pr) continue;
        const int pc = world / pr;
        if (pr > nr || pc > nc) continue;
        const double score = std::fabs(std::log(((double)nr / (double)pr) / ((double)nc / (double)pc)));
        if (!have || score < best) {
            have = true;
            best = score;
            *pr_out = pr;
            *pc_out = pc;
        }
    }
    return have;
}

// cuts of one exchange for `rank`: runs of windows that are neighbours in the SENDER's own-order array become one message
void messages_of(const td_seam_plan& p, int rank, int64_t wb, std::vector<td_seam_msg>* sends, std::vector<td_seam_msg>* recvs) {
    const int nc = p.nc();
    for (int d = 0; d < p.world; ++d) {
        if (d == rank) continue;
        const std::vector<Win> wins = p.crossing(rank, d);
        for (size_t i = 0; i < wins.size();) {
            const int li = p.local[rank][wins[i].first * nc + wins[i].second];
            size_t j = i + 1;
            while (j < wins.size() && p.local[rank][wins[j].first * nc + wins[j].second] == li + (int)(j - i)) ++j;
            sends->push_back(td_seam_msg{d, 0, (int64_t)li * wb, (int64_t)(j - i) * wb});
            i = j;
        }
    }
    int64_t slot = 0;
    for (int s = 0; s < p.world; ++s) {
        if (s == rank) continue;
        const std::vector<Win> wins = p.crossing(s, rank);
        for (size_t i = 0; i < wins.size();) {
            const int li = p.local[s][wins[i].first * nc + wins[i].second];
            size_t j = i + 1;
            while (j < wins.size() && p.local[s][wins[j].first * nc + wins[j].second] == li + (int)(j - i)) ++j;
            recvs->push_back(td_seam_msg{s, 0, slot * wb, (int64_t)(j - i) * wb});
            slot += (int64_t)(j - i);
            i = j;
        }
    }
}

int rccl_fail(const char* what, ncclResult_t r) {
    return fail(TD_SEAM_ERR_RCCL, std::string(what) + ": " + ncclGetErrorString(r));
}

}  // namespace

extern "C" {

const char* td_seam_last_error(void) { return g_err.c_str(); }

int td_seam_plan_create(int H, int W, int tile, int stride, int world, int extended, td_seam_plan** out) {
    if (!out) return fail(TD_SEAM_ERR_ARG, "td_seam_plan_create: plan pointer is NULL");
    *out = nullptr;
    if (H <= 0 || W <= 0 || tile <= 0 || stride < 0 || world <= 0)
        return fail(TD_SEAM_ERR_ARG, "td_seam_plan_create: H, W, tile and world must be positive, stride >= 0");
    if (stride == 0) stride = tile / 2;
    if (stride <= 0) stride = 1;
    td_seam_plan* p = new td_seam_plan();
    p->H = H; p->W = W; p->tile = tile; p->stride = stride; p->world = world; p->extended = extended ? 1 : 0;
    p->hs = starts_of(H, tile, stride);
    p->ws = starts_of(W, tile, stride);
    const int nr = p->nr(), nc = p->nc();
    if (!mesh_shape(world, nr, nc, &p->pr, &p->pc)) {
        char buf[160];
        std::snprintf(buf, sizeof buf, "cannot place %d ranks on a %dx%d window grid", world, nr, nc);
        delete p;
        return fail(TD_SEAM_ERR_ARG, buf);
    }
    for (int i = 0; i <= p->pr; ++i) p->row_cuts.push_back((int)(((int64_t)i * nr) / p->pr));
    for (int i = 0; i <= p->pc; ++i) p->col_cuts.push_back((int)(((int64_t)i * nc) / p->pc));
    p->owner.assign((size_t)nr * nc, -1);
    p->own.resize(world);
    p->local.assign(world, std::vector<int>((size_t)nr * nc, -1));
    for (int br = 0; br < p->pr; ++br)
        for (int bc = 0; bc < p->pc; ++bc) {
            const int r = br * p->pc + bc;
            for (int ic = p->row_cuts[br]; ic < p->row_cuts[br + 1]; ++ic)
                for (int jc = p->col_cuts[bc]; jc < p->col_cuts[bc + 1]; ++jc) {
                    p->owner[(size_t)ic * nc + jc] = r;
                    p->local[r][(size_t)ic * nc + jc] = (int)p->own[r].size();
                    p->own[r].push_back({ic, jc});
                }
            int y0, y1, x0, x1;
            if (p->extended) {   // bounding box of the block's own windows
                y0 = p->hs[p->row_cuts[br]];
                x0 = p->ws[p->col_cuts[bc]];
                y1 = p->hs[p->row_cuts[br + 1] - 1] + tile;
                x1 = p->ws[p->col_cuts[bc + 1] - 1] + tile;
            } else {             // from the origin of the block's first window to the origin of the next block's first window
                y0 = br > 0 ? p->hs[p->row_cuts[br]] : 0;
                y1 = br + 1 < p->pr ? p->hs[p->row_cuts[br + 1]] : H;
                x0 = bc > 0 ? p->ws[p->col_cuts[bc]] : 0;
                x1 = bc + 1 < p->pc ? p->ws[p->col_cuts[bc + 1]] : W;
            }
            p->regions.push_back({y0, y1 < H ? y1 : H, x0, x1 < W ? x1 : W});
        }
    p->needed.resize(world);
    for (int r = 0; r < world; ++r) {
        const auto& g = p->regions[r];
        for (int ic = 0; ic < nr; ++ic)
            for (int jc = 0; jc < nc; ++jc)
                if (p->hs[ic] < g[1] && p->hs[ic] + tile > g[0] && p->ws[jc] < g[3] && p->ws[jc] + tile > g[2]) p->needed[r].push_back({ic, jc});
    }
    *out = p;
    return TD_SEAM_OK;
}

void td_seam_plan_destroy(td_seam_plan* plan) { delete plan; }

int td_seam_plan_mesh(const td_seam_plan* p, int32_t mesh[4]) {
    if (!p || !mesh) return fail(TD_SEAM_ERR_ARG, "td_seam_plan_mesh: NULL argument");
    mesh[0] = p->pr; mesh[1] = p->pc; mesh[2] = p->nr(); mesh[3] = p->nc();
    return TD_SEAM_OK;
}

int td_seam_plan_region(const td_seam_plan* p, int rank, int32_t region[4]) {
    if (!p || !region) return fail(TD_SEAM_ERR_ARG, "td_seam_plan_region: NULL argument");
    if (rank < 0 || rank >= p->world) return fail(TD_SEAM_ERR_ARG, "td_seam_plan_region: rank outside the plan's world");
    for (int k = 0; k < 4; ++k) region[k] = p->regions[rank][k];
    return TD_SEAM_OK;
}

int td_seam_plan_starts(const td_seam_plan* p, int axis, int32_t* out, int cap) {
    if (!p || (axis != 0 && axis != 1)) return fail(TD_SEAM_ERR_ARG, "td_seam_plan_starts: NULL plan or axis not 0 / 1");
    const std::vector<int>& s = axis == 0 ? p->hs : p->ws;
    for (int k = 0; out && k < cap && k < (int)s.size(); ++k) out[k] = s[k];
    return (int)s.size();
}

int td_seam_plan_windows(const td_seam_plan* p, int rank, int kind, int32_t* ij, int32_t* peer, int cap) {
    if (!p) return fail(TD_SEAM_ERR_ARG, "td_seam_plan_windows: NULL plan");
    if (rank < 0 || rank >= p->world) return fail(TD_SEAM_ERR_ARG, "td_seam_plan_windows: rank outside the plan's world");
    std::vector<Win> wins;
    std::vector<int> peers;
    if (kind == TD_SEAM_OWN) {
        wins = p->own[rank];
        peers.assign(wins.size(), rank);
    } else if (kind == TD_SEAM_NEEDED) {
        wins = p->needed[rank];
        for (const Win& w : wins) peers.push_back(p->owner_of(w));
    } else if (kind == TD_SEAM_SENDS || kind == TD_SEAM_RECVS) {
        for (int o = 0; o < p->world; ++o) {
            if (o == rank) continue;
            const std::vector<Win> c = kind == TD_SEAM_SENDS ? p->crossing(rank, o) : p->crossing(o, rank);
            wins.insert(wins.end(), c.begin(), c.end());
            peers.insert(peers.end(), c.size(), o);
        }
    } else {
        return fail(TD_SEAM_ERR_ARG, "td_seam_plan_windows: kind is not one of TD_SEAM_OWN / NEEDED / SENDS / RECVS");
    }
    for (int k = 0; k < cap && k < (int)wins.size(); ++k) {
        if (ij) { ij[2 * k] = wins[k].first; ij[2 * k + 1] = wins[k].second; }
        if (peer) peer[k] = peers[k];
    }
    return (int)wins.size();
}

int td_seam_plan_messages(const td_seam_plan* p, int rank, int64_t window_bytes, td_seam_msg* sends, int* n_sends, td_seam_msg* recvs, int* n_recvs,
                          int cap) {
    if (!p || !n_sends || !n_recvs) return fail(TD_SEAM_ERR_ARG, "td_seam_plan_messages: NULL argument");
    if (rank < 0 || rank >= p->world) return fail(TD_SEAM_ERR_ARG, "td_seam_plan_messages: rank outside the plan's world");
    if (window_bytes <= 0) return fail(TD_SEAM_ERR_ARG, "td_seam_plan_messages: window_bytes must be positive");
    std::vector<td_seam_msg> s, r;
    messages_of(*p, rank, window_bytes, &s, &r);
    for (int k = 0; sends && k < cap && k < (int)s.size(); ++k) sends[k] = s[k];
    for (int k = 0; recvs && k < cap && k < (int)r.size(); ++k) recvs[k] = r[k];
    *n_sends = (int)s.size();
    *n_recvs = (int)r.size();
    return TD_SEAM_OK;
}

int td_seam_unique_id(void* id128) {
    if (!id128) return fail(TD_SEAM_ERR_ARG, "td_seam_unique_id: NULL buffer");
    static_assert(sizeof(ncclUniqueId) == TD_SEAM_ID_BYTES, "id size");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    std::memcpy(id128, &id, sizeof id);
    return TD_SEAM_OK;
}

int td_seam_comm_create(int device, int world, int rank, const void* id128, td_seam_comm** out) {
    if (!out || !id128) return fail(TD_SEAM_ERR_ARG, "td_seam_comm_create: NULL argument");
    *out = nullptr;
    if (world <= 0 || rank < 0 || rank >= world) return fail(TD_SEAM_ERR_ARG, "td_seam_comm_create: need 0 <= rank < world");
    const hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail(TD_SEAM_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    const ncclResult_t r = ncclCommInitRank(&c, world, id, rank);
    if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    *out = new td_seam_comm{c, world, rank, device, true};
    return TD_SEAM_OK;
}

int td_seam_comm_adopt(void* nccl_comm, td_seam_comm** out) {
    if (!out || !nccl_comm) return fail(TD_SEAM_ERR_ARG, "td_seam_comm_adopt: NULL argument");
    *out = nullptr;
    ncclComm_t c = (ncclComm_t)nccl_comm;
    int world = 0, rank = 0, device = 0;
    ncclResult_t r = ncclCommCount(c, &world);
    if (r == ncclSuccess) r = ncclCommUserRank(c, &rank);
    if (r == ncclSuccess) r = ncclCommCuDevice(c, &device);
    if (r != ncclSuccess) return rccl_fail("td_seam_comm_adopt", r);
    *out = new td_seam_comm{c, world, rank, device, false};
    return TD_SEAM_OK;
}

void td_seam_comm_destroy(td_seam_comm* comm) {
    if (!comm) return;
    if (comm->owned && comm->comm) ncclCommDestroy(comm->comm);
    delete comm;
}

int td_seam_comm_info(const td_seam_comm* comm, int32_t info[3]) {
    if (!comm || !info) return fail(TD_SEAM_ERR_ARG, "td_seam_comm_info: NULL argument");
    info[0] = comm->world; info[1] = comm->rank; info[2] = comm->device;
    return TD_SEAM_OK;
}

int td_seam_exchange(td_seam_comm* comm, const void* send_base, const td_seam_msg* sends, int n_sends, void* recv_base, const td_seam_msg* recvs,
                     int n_recvs, void* hip_stream) {
    if (!comm) return fail(TD_SEAM_ERR_ARG, "td_seam_exchange: NULL communicator");
    if (n_sends < 0 || n_recvs < 0 || (n_sends && !sends) || (n_recvs && !recvs)) return fail(TD_SEAM_ERR_ARG, "td_seam_exchange: bad message lists");
    for (int pass = 0; pass < 2; ++pass) {
        const td_seam_msg* m = pass ? recvs : sends;
        for (int k = 0; k < (pass ? n_recvs : n_sends); ++k)
            if (m[k].peer < 0 || m[k].peer >= comm->world || m[k].bytes <= 0 || m[k].offset < 0)
                return fail(TD_SEAM_ERR_ARG, "td_seam_exchange: a message names a peer outside the communicator, or has no bytes / a negative offset");
    }
    if (n_sends + n_recvs == 0) return TD_SEAM_OK;
    hipStream_t st = (hipStream_t)hip_stream;
    // RCCL posts on the communicator's device: make it current for the group and put the caller's back afterwards
    int prev = comm->device;
    if (hipGetDevice(&prev) != hipSuccess) prev = comm->device;
    if (prev != comm->device && hipSetDevice(comm->device) != hipSuccess) return fail(TD_SEAM_ERR_HIP, "hipSetDevice to the communicator's device failed");
    struct Restore {
        int prev, cur;
        ~Restore() { if (prev != cur) (void)hipSetDevice(prev); }
    } restore{prev, comm->device};
    ncclResult_t r = ncclGroupStart();
    if (r != ncclSuccess) return rccl_fail("ncclGroupStart", r);
    ncclResult_t bad = ncclSuccess;
    for (int k = 0; k < n_sends && bad == ncclSuccess; ++k)
        bad = ncclSend((const char*)send_base + sends[k].offset, (size_t)sends[k].bytes, ncclInt8, sends[k].peer, comm->comm, st);
    for (int k = 0; k < n_recvs && bad == ncclSuccess; ++k)
        bad = ncclRecv((char*)recv_base + recvs[k].offset, (size_t)recvs[k].bytes, ncclInt8, recvs[k].peer, comm->comm, st);
    r = ncclGroupEnd();   // always closed, also after a failed post
    if (bad != ncclSuccess) return rccl_fail("ncclSend/ncclRecv", bad);
    if (r != ncclSuccess) return rccl_fail("ncclGroupEnd", r);
    return TD_SEAM_OK;
}

int td_seam_exchange_windows(td_seam_comm* comm, const td_seam_plan* plan, const void* my_tiles, void* recv_tiles, int64_t window_bytes,
                             void* hip_stream) {
    if (!comm || !plan) return fail(TD_SEAM_ERR_ARG, "td_seam_exchange_windows: NULL communicator or plan");
    if (plan->world != comm->world) return fail(TD_SEAM_ERR_ARG, "td_seam_exchange_windows: the plan's world differs from the communicator's");
    if (window_bytes <= 0) return fail(TD_SEAM_ERR_ARG, "td_seam_exchange_windows: window_bytes must be positive");
    std::vector<td_seam_msg> s, r;
    messages_of(*plan, comm->rank, window_bytes, &s, &r);
    if ((!s.empty() && !my_tiles) || (!r.empty() && !recv_tiles)) return fail(TD_SEAM_ERR_ARG, "td_seam_exchange_windows: NULL tile buffer");
    return td_seam_exchange(comm, my_tiles, s.data(), (int)s.size(), recv_tiles, r.data(), (int)r.size(), hip_stream);
}

}  // extern "C"
