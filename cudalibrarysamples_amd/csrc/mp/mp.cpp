// mp.cpp — cuTENSORMp on MI355X: one process per GPU, RCCL point-to-point exchange over xGMI.
//
// Serves the call sequence of cutensorMp/cutensorMp_contraction.cu (handle :470-471, distributed tensor
// descriptors :473-483, contraction / preference / plan :485-509, cutensorMpContract :537-538).  Built on the
// public single-GPU ABI (include/cutensor.h), HIP and RCCL.
//
// Algorithm ("owner computes, gather what you need"):
//   1. Rank r owns one block ("cell") of C / D.  Its range in every mode of C fixes the box of A and of B it needs:
//      the same range in the modes an operand shares with C, the full extent in the contracted modes.
//   2. Every rank intersects that box with the cells of A and B.  A non-empty intersection is one transfer
//      owner -> r of exactly that sub-box: the owner packs it (a strided identity cutensorPermute) unless it is its
//      whole contiguous block, all transfers of a call travel in ONE ncclGroup of send/recv pairs, and the receiver
//      unpacks into a dense staging tensor.  Both sides derive the same transfer list from the descriptors, so no
//      metadata is exchanged.  An operand whose needed box lies inside the rank's own block is used in place.
//   3. One cutensorContract per rank: staged (or in-place) views of A and B, D written straight into the rank's
//      own block — the result needs no scatter and beta * C needs no exchange.
// The local contraction therefore always sees dense, packed operands (the GETT engine's fast layouts); the price
// is that a contracted mode distributed over ranks is gathered rather than reduced.
//
// Second algorithm ("stationary inputs, reduce the result") for the opposite shape — A and B distributed identically
// along contracted modes only, e.g. the headline einsum with K cut over the ranks: every rank contracts its own K
// slice in place (no operand moves at all) into a full-size partial of C, and ONE RCCL collective finishes the job:
// ncclAllReduce when C is replicated, ncclReduceScatter (destination blocks packed into equal slots) when C is
// distributed.  beta * C enters the sum exactly once (rank 0 for a replicated C, the owner's block otherwise).
// Every rank picks the algorithm from the same global byte counts, so the choice needs no communication.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cutensor.h>
#include <cutensorMp.h>

#include "../host/api_guard.hpp"

namespace {

size_t elem_size(hipDataType t) {
    switch (t) {
        case HIP_R_16F: case HIP_R_16BF: return 2;
        case HIP_R_32F: return 4;
        case HIP_R_64F: case HIP_C_32F: return 8;
        case HIP_C_64F: return 16;
        default: return 0;
    }
}
bool is_complex(hipDataType t) { return t == HIP_C_32F || t == HIP_C_64F; }
hipDataType real_type(hipDataType t) { return t == HIP_C_32F ? HIP_R_32F : t == HIP_C_64F ? HIP_R_64F : t; }

int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// ---- exchange layer ------------------------------------------------------------------------------------------
class Transport {
public:
    virtual ~Transport() {}
    virtual int rank() const = 0;
    virtual int size() const = 0;
    virtual bool begin() = 0;
    virtual bool send(const void* buf, size_t bytes, int peer, hipStream_t s) = 0;
    virtual bool recv(void* buf, size_t bytes, int peer, hipStream_t s) = 0;
    virtual bool end(hipStream_t s) = 0;
    // sums of `count` elements of the real type `t`; scratch: `count` elements, used by transports that need it
    virtual bool all_reduce(void* buf, size_t count, hipDataType t, void* scratch, hipStream_t s) = 0;
    virtual bool reduce_scatter(const void* send, void* recv, size_t countPerRank, hipDataType t, hipStream_t s) = 0;
    virtual bool needs_scratch() const { return false; }
};

ncclDataType_t nccl_type(hipDataType t) {
    switch (t) {
        case HIP_R_16F: return ncclFloat16;
        case HIP_R_16BF: return ncclBfloat16;
        case HIP_R_64F: return ncclFloat64;
        default: return ncclFloat32;
    }
}

class RcclTransport : public Transport {
public:
    RcclTransport(ncclComm_t c, int r, int n) : comm_(c), rank_(r), size_(n) {}
    int rank() const override { return rank_; }
    int size() const override { return size_; }
    bool begin() override { return ncclGroupStart() == ncclSuccess; }
    bool send(const void* buf, size_t bytes, int peer, hipStream_t s) override {
        return ncclSend(buf, bytes, ncclInt8, peer, comm_, s) == ncclSuccess;
    }
    bool recv(void* buf, size_t bytes, int peer, hipStream_t s) override {
        return ncclRecv(buf, bytes, ncclInt8, peer, comm_, s) == ncclSuccess;
    }
    bool end(hipStream_t) override { return ncclGroupEnd() == ncclSuccess; }
    bool all_reduce(void* buf, size_t count, hipDataType t, void*, hipStream_t s) override {
        return ncclAllReduce(buf, buf, count, nccl_type(t), ncclSum, comm_, s) == ncclSuccess;
    }
    bool reduce_scatter(const void* send, void* recv, size_t countPerRank, hipDataType t, hipStream_t s) override {
        return ncclReduceScatter(send, recv, countPerRank, nccl_type(t), ncclSum, comm_, s) == ncclSuccess;
    }
private:
    ncclComm_t comm_;
    int rank_, size_;
};

// Several ranks as threads of one process on one GPU (see cutensorMp.h).  A send posts {buffer, "data ready" event};
// the matching recv copies device-to-device on the receiver's stream and posts a "consumed" event the sender's
// stream then waits for — the same completion contract as an RCCL send/recv pair.
struct LocalWorld {
    struct Msg { const void* buf; size_t bytes; hipEvent_t ready; hipEvent_t consumed; bool done; };
    int nranks;
    std::mutex mu;
    std::condition_variable cv;
    std::map<std::tuple<int, int, uint64_t>, Msg> box;   // (src, dst, sequence number of the pair)
    std::vector<uint64_t> sendSeq, recvSeq;              // [src * nranks + dst]
    std::vector<const void*> published;                  // collectives: every rank's buffer
    int arrived = 0;
    uint64_t generation = 0;
    explicit LocalWorld(int n) : nranks(n), sendSeq((size_t)n * n, 0), recvSeq((size_t)n * n, 0), published((size_t)n, nullptr) {}
    // host barrier of the rank threads; false when a rank never arrives
    bool barrier(int timeoutS) {
        std::unique_lock<std::mutex> lock(mu);
        const uint64_t gen = generation;
        if (++arrived == nranks) { arrived = 0; ++generation; cv.notify_all(); return true; }
        return cv.wait_for(lock, std::chrono::seconds(timeoutS), [&] { return generation != gen; });
    }
};

class LocalTransport : public Transport {
public:
    LocalTransport(LocalWorld* w, int r) : w_(w), rank_(r) {}
    int rank() const override { return rank_; }
    int size() const override { return w_->nranks; }
    bool begin() override { sent_.clear(); pending_.clear(); return true; }
    bool send(const void* buf, size_t bytes, int peer, hipStream_t s) override {
        LocalWorld::Msg m{buf, bytes, nullptr, nullptr, false};
        if (hipEventCreateWithFlags(&m.ready, hipEventDisableTiming) != hipSuccess) return false;
        if (hipEventRecord(m.ready, s) != hipSuccess) return false;
        std::lock_guard<std::mutex> lock(w_->mu);
        const uint64_t seq = w_->sendSeq[(size_t)rank_ * w_->nranks + peer]++;
        w_->box[std::make_tuple(rank_, peer, seq)] = m;
        sent_.push_back(std::make_tuple(rank_, peer, seq));
        w_->cv.notify_all();
        return true;
    }
    bool recv(void* buf, size_t bytes, int peer, hipStream_t) override {
        pending_.push_back(Pending{buf, bytes, peer});
        return true;
    }
    bool end(hipStream_t s) override {
        bool ok = true;
        for (const Pending& p : pending_) {
            std::unique_lock<std::mutex> lock(w_->mu);
            const uint64_t seq = w_->recvSeq[(size_t)peer_index(p.peer)]++;
            const auto key = std::make_tuple(p.peer, rank_, seq);
            // a peer that never shows up is a usage error (not every rank entered the call): fail, do not hang
            if (!w_->cv.wait_for(lock, std::chrono::seconds(kPeerTimeoutS), [&] { return w_->box.count(key) > 0; })) return false;
            LocalWorld::Msg& m = w_->box[key];
            ok = ok && m.bytes == p.bytes;
            ok = ok && hipStreamWaitEvent(s, m.ready, 0) == hipSuccess;
            ok = ok && hipMemcpyAsync(p.buf, m.buf, std::min(p.bytes, m.bytes), hipMemcpyDeviceToDevice, s) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&m.consumed, hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventRecord(m.consumed, s) == hipSuccess;
            m.done = true;
            w_->cv.notify_all();
        }
        for (const auto& key : sent_) {
            std::unique_lock<std::mutex> lock(w_->mu);
            if (!w_->cv.wait_for(lock, std::chrono::seconds(kPeerTimeoutS), [&] { return w_->box[key].done; })) return false;
            LocalWorld::Msg m = w_->box[key];
            w_->box.erase(key);
            lock.unlock();
            if (m.consumed) { ok = ok && hipStreamWaitEvent(s, m.consumed, 0) == hipSuccess; (void)hipEventDestroy(m.consumed); }
            (void)hipEventDestroy(m.ready);
        }
        sent_.clear(); pending_.clear();
        return ok;
    }
    // Collectives: stream-synchronous on purpose (this transport exists for verification, not speed).  Every rank
    // publishes its buffer, sums all of them slot-wise with the engine's own element-wise ADD, and nobody touches a
    // published buffer again before every rank has finished reading it.
    void set_engine(cutensorHandle_t h) { engine_ = h; }
    bool needs_scratch() const override { return true; }
    bool all_reduce(void* buf, size_t count, hipDataType t, void* scratch, hipStream_t s) override {
        if (!publish(buf, s)) return false;
        bool ok = sum_slots(scratch, 0, count, t, s);
        ok = hipStreamSynchronize(s) == hipSuccess && ok;
        ok = w_->barrier(kPeerTimeoutS) && ok;      // every rank has read every buffer
        ok = ok && hipMemcpyAsync(buf, scratch, count * real_size(t), hipMemcpyDeviceToDevice, s) == hipSuccess;
        return ok;
    }
    bool reduce_scatter(const void* send, void* recv, size_t countPerRank, hipDataType t, hipStream_t s) override {
        if (!publish(send, s)) return false;
        bool ok = sum_slots(recv, (size_t)rank_ * countPerRank, countPerRank, t, s);
        ok = hipStreamSynchronize(s) == hipSuccess && ok;
        return w_->barrier(kPeerTimeoutS) && ok;
    }
private:
    static constexpr int kPeerTimeoutS = 60;
    static size_t real_size(hipDataType t) { return t == HIP_R_64F ? 8 : (t == HIP_R_32F ? 4 : 2); }
    bool publish(const void* buf, hipStream_t s) {
        if (hipStreamSynchronize(s) != hipSuccess) return false;
        { std::lock_guard<std::mutex> lock(w_->mu); w_->published[(size_t)rank_] = buf; }
        return w_->barrier(kPeerTimeoutS);
    }
    // dst[0..count) = sum over ranks q of published[q][offset .. offset + count)
    bool sum_slots(void* dst, size_t offset, size_t count, hipDataType t, hipStream_t s) {
        const size_t es = real_size(t);
        const int64_t ext[1] = {(int64_t)count};
        const int32_t lab[1] = {0};
        cutensorTensorDescriptor_t d = nullptr;
        cutensorOperationDescriptor_t op = nullptr;
        cutensorPlan_t plan = nullptr;
        cutensorStatus_t st = cutensorCreateTensorDescriptor(engine_, &d, 1, ext, nullptr, t, (uint32_t)es);
        if (st == CUTENSOR_STATUS_SUCCESS)
            st = cutensorCreateElementwiseBinary(engine_, &op, d, lab, CUTENSOR_OP_IDENTITY, d, lab, CUTENSOR_OP_IDENTITY, d, lab, CUTENSOR_OP_ADD,
                                                 t == HIP_R_64F ? CUTENSOR_COMPUTE_DESC_64F : CUTENSOR_COMPUTE_DESC_32F);
        if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreatePlan(engine_, &plan, op, nullptr, 0);
        const float onef = 1.f;
        const double oned = 1.0;
        const void* one = t == HIP_R_64F ? static_cast<const void*>(&oned) : static_cast<const void*>(&onef);
        bool ok = st == CUTENSOR_STATUS_SUCCESS;
        for (int q = 0; q < w_->nranks && ok; ++q) {
            const char* src = static_cast<const char*>(w_->published[(size_t)q]) + offset * es;
            if (q == 0) ok = hipMemcpyAsync(dst, src, count * es, hipMemcpyDeviceToDevice, s) == hipSuccess;
            else ok = cutensorElementwiseBinaryExecute(engine_, plan, one, src, one, dst, dst, s) == CUTENSOR_STATUS_SUCCESS;
        }
        cutensorDestroyPlan(plan);
        cutensorDestroyOperationDescriptor(op);
        cutensorDestroyTensorDescriptor(d);
        return ok;
    }
    cutensorHandle_t engine_ = nullptr;
    struct Pending { void* buf; size_t bytes; int peer; };
    size_t peer_index(int peer) const { return (size_t)peer * w_->nranks + rank_; }
    LocalWorld* w_;
    int rank_;
    std::vector<std::tuple<int, int, uint64_t>> sent_;
    std::vector<Pending> pending_;
};

// ---- distributed tensor geometry --------------------------------------------------------------------------------
struct MpTensor {
    uint32_t n = 0;
    hipDataType dtype = HIP_R_32F;
    std::vector<int64_t> extent, p, bs, elemStride, cellStride;
    int64_t numCells = 1;
    std::vector<int32_t> owner;     // rank of each cell; empty = replicated on every rank
    bool packed = true;             // elemStride is the packed layout over bs
    bool replicated() const { return owner.empty(); }
};

struct Box {
    std::vector<int64_t> lo, hi;
    bool empty() const {
        for (size_t i = 0; i < lo.size(); ++i) if (hi[i] <= lo[i]) return true;
        return false;
    }
    int64_t volume() const {
        int64_t v = 1;
        for (size_t i = 0; i < lo.size(); ++i) v *= std::max<int64_t>(0, hi[i] - lo[i]);
        return v;
    }
    bool operator==(const Box& o) const { return lo == o.lo && hi == o.hi; }
};

Box cell_box(const MpTensor& t, int64_t cell) {
    Box b; b.lo.resize(t.n); b.hi.resize(t.n);
    for (uint32_t i = 0; i < t.n; ++i) {
        const int64_t c = t.replicated() ? 0 : (cell / t.cellStride[i]) % t.p[i];
        b.lo[i] = c * t.bs[i];
        b.hi[i] = std::min(t.extent[i], b.lo[i] + t.bs[i]);
    }
    return b;
}
Box intersect(const Box& a, const Box& b) {
    Box r; r.lo.resize(a.lo.size()); r.hi.resize(a.lo.size());
    for (size_t i = 0; i < a.lo.size(); ++i) { r.lo[i] = std::max(a.lo[i], b.lo[i]); r.hi[i] = std::min(a.hi[i], b.hi[i]); }
    return r;
}
int64_t cell_of_rank(const MpTensor& t, int rank) {
    if (t.replicated()) return 0;
    for (int64_t c = 0; c < t.numCells; ++c) if (t.owner[c] == rank) return c;
    return -1;
}
int find_label(const std::vector<int32_t>& v, int32_t l) {
    for (size_t i = 0; i < v.size(); ++i) if (v[i] == l) return (int)i;
    return -1;
}

// ---- strided copies through the single-GPU ABI ------------------------------------------------------------------
struct CMode { int64_t extent, sSrc, sDst; };
struct CopyOp {
    cutensorPlan_t plan = nullptr;
    std::vector<std::pair<int64_t, int64_t>> launches;   // (src, dst) element offsets added to the base pointers
};

cutensorStatus_t try_copy_plan(cutensorHandle_t h, hipDataType dtype, const std::vector<CMode>& modes, uint32_t alignSrc, uint32_t alignDst,
                               cutensorPlan_t* plan) {
    // complex data is copied as pairs of reals: a leading stride-1 mode of extent 2
    const hipDataType rt = real_type(dtype);
    const int64_t scale = is_complex(dtype) ? 2 : 1;
    std::vector<int64_t> ext, sS, sD;
    std::vector<int32_t> lab;
    if (scale == 2) { ext.push_back(2); sS.push_back(1); sD.push_back(1); lab.push_back(0); }
    for (const CMode& m : modes) { ext.push_back(m.extent); sS.push_back(m.sSrc * scale); sD.push_back(m.sDst * scale); lab.push_back((int32_t)lab.size() + 1); }
    cutensorTensorDescriptor_t dS = nullptr, dD = nullptr;
    cutensorOperationDescriptor_t op = nullptr;
    cutensorStatus_t st = cutensorCreateTensorDescriptor(h, &dS, (uint32_t)ext.size(), ext.data(), sS.data(), rt, alignSrc);
    if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreateTensorDescriptor(h, &dD, (uint32_t)ext.size(), ext.data(), sD.data(), rt, alignDst);
    if (st == CUTENSOR_STATUS_SUCCESS)
        st = cutensorCreatePermutation(h, &op, dS, lab.data(), CUTENSOR_OP_IDENTITY, dD, lab.data(),
                                       rt == HIP_R_64F ? CUTENSOR_COMPUTE_DESC_64F : CUTENSOR_COMPUTE_DESC_32F);
    if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreatePlan(h, plan, op, nullptr, 0);
    cutensorDestroyOperationDescriptor(op);
    cutensorDestroyTensorDescriptor(dS);
    cutensorDestroyTensorDescriptor(dD);
    return st;
}

// Largest power of two <= 256 dividing every byte offset: what the copy may promise about its pointers (the bases it is
// added to — hipMalloc'ed user blocks, 256-byte-aligned workspace regions — are 256-byte aligned).
uint32_t common_alignment(const std::vector<std::pair<int64_t, int64_t>>& launches, bool second, int64_t es) {
    uint32_t a = 256;
    for (const auto& l : launches) {
        const int64_t bytes = (second ? l.second : l.first) * es;
        while (a > 1 && (bytes % a) != 0) a >>= 1;
    }
    return std::max<uint32_t>(a, (uint32_t)std::min<int64_t>(es, 16));
}

// Builds the copy of a box that starts srcBase / dstBase elements into its buffers; when the element-wise planner rejects
// the view (too many unfusable modes) the smallest group is peeled into a host loop over launches of the remaining view.
cutensorStatus_t make_copy(cutensorHandle_t h, hipDataType dtype, std::vector<CMode> modes, CopyOp& op, int64_t srcBase = 0, int64_t dstBase = 0) {
    // drop unit modes, fuse modes that are contiguous in both tensors
    std::vector<CMode> g;
    for (const CMode& m : modes) {
        if (m.extent == 1) continue;
        if (!g.empty() && m.sSrc == g.back().sSrc * g.back().extent && m.sDst == g.back().sDst * g.back().extent) { g.back().extent *= m.extent; continue; }
        g.push_back(m);
    }
    const int64_t es = (int64_t)elem_size(dtype);
    op.launches.assign(1, std::make_pair(srcBase, dstBase));
    for (;;) {
        cutensorStatus_t st = try_copy_plan(h, dtype, g, common_alignment(op.launches, false, es), common_alignment(op.launches, true, es), &op.plan);
        if (st == CUTENSOR_STATUS_SUCCESS) return st;
        if (st != CUTENSOR_STATUS_NOT_SUPPORTED || g.empty()) return st;
        size_t k = 0;
        for (size_t i = 1; i < g.size(); ++i) if (g[i].extent < g[k].extent) k = i;
        const CMode m = g[k];
        g.erase(g.begin() + (long)k);
        if ((int64_t)op.launches.size() * m.extent > 4096) return CUTENSOR_STATUS_NOT_SUPPORTED;
        std::vector<std::pair<int64_t, int64_t>> nl;
        for (const auto& l : op.launches)
            for (int64_t j = 0; j < m.extent; ++j) nl.push_back(std::make_pair(l.first + j * m.sSrc, l.second + j * m.sDst));
        op.launches.swap(nl);
    }
}

std::vector<int64_t> packed_strides(const std::vector<int64_t>& ext) {
    std::vector<int64_t> s(ext.size());
    int64_t run = 1;
    for (size_t i = 0; i < ext.size(); ++i) { s[i] = run; run *= ext[i]; }
    return s;
}
std::vector<int64_t> sizes(const Box& b) {
    std::vector<int64_t> s(b.lo.size());
    for (size_t i = 0; i < s.size(); ++i) s[i] = b.hi[i] - b.lo[i];
    return s;
}

// ---- plan ---------------------------------------------------------------------------------------------------
enum Where { IN_USER = 0, IN_SEND = 1, IN_RECV = 2, IN_STAGE = 3 };

struct Transfer {            // one sub-box of operand `tensor` (0 = A, 1 = B) moving from rank `src` to rank `dst`
    int tensor, src, dst;
    Box box;
    int64_t bytes;
    // sender side
    bool direct = false;     // sent straight from the user's block (whole, contiguous)
    int64_t sendOff = 0;     // byte offset in the send region
    CopyOp pack;
    // receiver side
    bool inPlace = false;    // received straight into the staging tensor (single source covering the whole box)
    int64_t recvOff = 0;     // byte offset in the receive region
    CopyOp unpack;
};

struct OperandPlan {
    bool staged = false;             // false: the local contraction reads the user's block in place
    Box need;
    int64_t viewOff = 0;             // element offset of the view inside the user's block (in-place only)
    std::vector<int64_t> viewStride;
    int64_t stageOff = 0, stageBytes = 0;
    std::vector<CopyOp> localCopies; // own block -> staging
};

// "stationary inputs, reduce the result" (see the header of this file)
struct ReducePlan {
    bool replicatedC = false;
    bool direct = false;             // replicated C, packed: contract and all-reduce in the user's D
    int64_t stageOff = 0, stageBytes = 0;     // the full-size partial of C
    CopyOp cIn;                      // beta != 0: the rank's (block of) C -> partial
    CopyOp out;                      // replicated, not direct: reduced partial -> D
    bool packDirect = false;         // the destination blocks are already equal contiguous slots of the partial
    std::vector<CopyOp> pack;        // partial -> slot q of the send region
    int64_t sendOff = 0, sendBytes = 0, slotElems = 0;
    bool recvDirect = false;         // the rank's D block is a packed slot: reduce-scatter straight into it
    int64_t recvOff = 0, recvBytes = 0;
    CopyOp unpack;                   // received slot -> D block
    int64_t scratchOff = 0, scratchBytes = 0;
};

}  // namespace

struct cutensorMpHandle {
    Transport* transport = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    cutensorHandle_t h = nullptr;
    ~cutensorMpHandle() { delete transport; if (h) cutensorDestroy(h); }
};
struct cutensorMpTensorDescriptor { MpTensor t; };
struct cutensorMpOperationDescriptor {
    MpTensor A, B, C;
    std::vector<int32_t> mA, mB, mC;
    cutensorOperator_t opA, opB, opC;
    cutensorComputeDescriptor_t compute;
};
struct cutensorMpPlanPreference { cutensorMpAlgo_t algo; uint64_t devLimit, hostLimit; };
struct cutensorMpPlan {
    cutensorMpOperationDescriptor desc;
    int rank = 0, nranks = 1;
    std::vector<Transfer> sends, recvs;     // in issue order (tensor-major, then peer / cell) — identical on both sides
    OperandPlan in[2];
    bool compute = false;                   // this rank owns a non-empty block of D
    cutensorPlan_t contraction = nullptr;
    uint64_t contractionWs = 0;
    int64_t sendBytes = 0, recvBytes = 0, stageBytes = 0;
    uint64_t requiredDevice = 0;
    bool reduce = false;                    // algorithm: false = gather operands, true = reduce the result
    ReducePlan red;
    int64_t gatherTotal = 0, reduceTotal = 0;   // bytes on the wire over all ranks, per algorithm (the selection rule)
    ~cutensorMpPlan() {
        cutensorDestroyPlan(red.cIn.plan);
        cutensorDestroyPlan(red.out.plan);
        cutensorDestroyPlan(red.unpack.plan);
        for (CopyOp& c : red.pack) cutensorDestroyPlan(c.plan);
        for (Transfer& t : sends) cutensorDestroyPlan(t.pack.plan);
        for (Transfer& t : recvs) cutensorDestroyPlan(t.unpack.plan);
        for (OperandPlan& o : in) for (CopyOp& c : o.localCopies) cutensorDestroyPlan(c.plan);
        cutensorDestroyPlan(contraction);
    }
};

namespace {

Box needed_box(const cutensorMpOperationDescriptor& d, int tensor, const Box& cBox) {
    const MpTensor& x = tensor == 0 ? d.A : d.B;
    const std::vector<int32_t>& mx = tensor == 0 ? d.mA : d.mB;
    Box b; b.lo.resize(x.n); b.hi.resize(x.n);
    for (uint32_t i = 0; i < x.n; ++i) {
        const int j = find_label(d.mC, mx[i]);
        if (j >= 0) { b.lo[i] = cBox.lo[j]; b.hi[i] = cBox.hi[j]; }
        else { b.lo[i] = 0; b.hi[i] = x.extent[i]; }
    }
    return b;
}

// modes of a copy: box `b` read at `srcStride` relative to `srcOrigin`, written at `dstStride` relative to `dstOrigin`
std::vector<CMode> copy_modes(const Box& b, const std::vector<int64_t>& srcStride, const std::vector<int64_t>& dstStride) {
    std::vector<CMode> m;
    for (size_t i = 0; i < b.lo.size(); ++i) m.push_back(CMode{b.hi[i] - b.lo[i], srcStride[i], dstStride[i]});
    return m;
}
int64_t box_offset(const Box& b, const Box& origin, const std::vector<int64_t>& stride) {
    int64_t off = 0;
    for (size_t i = 0; i < b.lo.size(); ++i) off += (b.lo[i] - origin.lo[i]) * stride[i];
    return off;
}

cutensorStatus_t run_copy(cutensorHandle_t h, const CopyOp& c, hipDataType dtype, const void* src, void* dst, hipStream_t s) {
    const float onef = 1.f;
    const double oned = 1.0;
    const void* one = real_type(dtype) == HIP_R_64F ? static_cast<const void*>(&oned) : static_cast<const void*>(&onef);
    const int64_t es = (int64_t)elem_size(dtype);
    for (const auto& l : c.launches) {
        cutensorStatus_t st = cutensorPermute(h, c.plan, one, static_cast<const char*>(src) + l.first * es,
                                              static_cast<char*>(dst) + l.second * es, s);
        if (st != CUTENSOR_STATUS_SUCCESS) return st;
    }
    return CUTENSOR_STATUS_SUCCESS;
}

uint32_t alignment_of(int64_t byteOffset) {
    uint32_t a = 256;
    while (a > 1 && (byteOffset % a) != 0) a >>= 1;
    return a;
}


// ---- algorithm selection and the reduce plan ---------------------------------------------------------------------
// Bytes every rank would receive under the gather algorithm, summed over ranks.
int64_t gather_traffic(const cutensorMpOperationDescriptor& d, const std::vector<Box>& cBox, int world, int64_t es) {
    int64_t total = 0;
    const MpTensor* X[2] = {&d.A, &d.B};
    for (int k = 0; k < 2; ++k) {
        const MpTensor& x = *X[k];
        if (x.replicated()) continue;
        for (int q = 0; q < world; ++q) {
            if (cBox[(size_t)q].empty()) continue;
            const Box need = needed_box(d, k, cBox[(size_t)q]);
            for (int64_t c = 0; c < x.numCells; ++c)
                if (x.owner[(size_t)c] != q) total += intersect(need, cell_box(x, c)).volume() * es;
        }
    }
    return total;
}

// The reduce algorithm applies when A and B are cut along contracted modes only and every rank holds the same
// non-empty K range of both.
bool reduce_applicable(const cutensorMpOperationDescriptor& d, int world) {
    if (d.A.replicated() || d.B.replicated() || world < 2) return false;
    const MpTensor* X[2] = {&d.A, &d.B};
    const std::vector<int32_t>* M[2] = {&d.mA, &d.mB};
    for (int k = 0; k < 2; ++k)
        for (uint32_t i = 0; i < X[k]->n; ++i) {
            if (X[k]->p[i] == 1) continue;
            if (find_label(d.mC, (*M[k])[i]) >= 0) return false;            // a free / batch mode is cut
            if (find_label(*M[1 - k], (*M[k])[i]) < 0) return false;
        }
    for (int r = 0; r < world; ++r) {
        const Box a = cell_box(d.A, cell_of_rank(d.A, r)), b = cell_box(d.B, cell_of_rank(d.B, r));
        if (a.empty() || b.empty()) return false;
        for (uint32_t i = 0; i < d.A.n; ++i) {
            const int j = find_label(d.mB, d.mA[i]);
            if (j >= 0 && (a.lo[i] != b.lo[(size_t)j] || a.hi[i] != b.hi[(size_t)j])) return false;
        }
    }
    return true;
}

// a box that is one contiguous run of a packed tensor of extents `full`
bool contiguous_in(const Box& b, const std::vector<int64_t>& full) {
    bool partialSeen = false;
    for (size_t i = 0; i < full.size(); ++i) {
        const int64_t sz = b.hi[i] - b.lo[i];
        if (partialSeen && sz != 1) return false;
        if (sz != full[i]) partialSeen = true;
    }
    return true;
}

cutensorStatus_t build_reduce_plan(cutensorMpHandle* handle, cutensorMpPlan* pl, const std::vector<Box>& cBox, uint64_t devLimit) {
    const cutensorMpOperationDescriptor& d = pl->desc;
    ReducePlan& R = pl->red;
    cutensorHandle_t h = handle->h;
    const int me = pl->rank, world = pl->nranks;
    const int64_t es = (int64_t)elem_size(d.C.dtype);
    const std::vector<int64_t> full = d.C.extent;
    const std::vector<int64_t> pStride = packed_strides(full);
    Box whole; whole.lo.assign(d.C.n, 0); whole.hi = full;
    const int64_t cElems = whole.volume();
    const Box myC = cBox[(size_t)me];
    cutensorStatus_t st = CUTENSOR_STATUS_SUCCESS;
    int64_t off = 0;
    auto region = [&](int64_t bytes, int64_t& at) { at = off; off += round_up(bytes, 256); };

    R.replicatedC = d.C.replicated();
    if (R.replicatedC) {
        R.direct = d.C.packed && d.C.bs == d.C.extent;
        if (!R.direct) {
            R.stageBytes = cElems * es;
            region(R.stageBytes, R.stageOff);
            st = make_copy(h, d.C.dtype, copy_modes(whole, d.C.elemStride, pStride), R.cIn);
            if (st == CUTENSOR_STATUS_SUCCESS) st = make_copy(h, d.C.dtype, copy_modes(whole, pStride, d.C.elemStride), R.out);
            if (st != CUTENSOR_STATUS_SUCCESS) return st;
        }
        if (handle->transport->needs_scratch()) { R.scratchBytes = cElems * es; region(R.scratchBytes, R.scratchOff); }
    } else {
        R.stageBytes = cElems * es;
        region(R.stageBytes, R.stageOff);
        if (!myC.empty()) {
            st = make_copy(h, d.C.dtype, copy_modes(myC, d.C.elemStride, pStride), R.cIn, 0, box_offset(myC, whole, pStride));
            if (st != CUTENSOR_STATUS_SUCCESS) return st;
        }
        for (int q = 0; q < world; ++q) R.slotElems = std::max(R.slotElems, cBox[(size_t)q].volume());
        R.slotElems = std::max<int64_t>(R.slotElems, 1);
        R.packDirect = true;
        for (int q = 0; q < world; ++q) {
            const Box& b = cBox[(size_t)q];
            if (b.empty() || b.volume() != R.slotElems || !contiguous_in(b, full) || box_offset(b, whole, pStride) != q * R.slotElems)
                R.packDirect = false;
        }
        if (!R.packDirect) {
            R.sendBytes = (int64_t)world * R.slotElems * es;
            region(R.sendBytes, R.sendOff);
            R.pack.resize((size_t)world);
            for (int q = 0; q < world; ++q) {
                const Box& b = cBox[(size_t)q];
                if (b.empty()) continue;
                st = make_copy(h, d.C.dtype, copy_modes(b, pStride, packed_strides(sizes(b))), R.pack[(size_t)q],
                               box_offset(b, whole, pStride), (int64_t)q * R.slotElems);
                if (st != CUTENSOR_STATUS_SUCCESS) return st;
            }
        }
        R.recvDirect = !myC.empty() && d.C.packed && sizes(myC) == d.C.bs && myC.volume() == R.slotElems;
        if (!R.recvDirect) {
            R.recvBytes = R.slotElems * es;
            region(R.recvBytes, R.recvOff);
            if (!myC.empty()) {
                st = make_copy(h, d.C.dtype, copy_modes(myC, packed_strides(sizes(myC)), d.C.elemStride), R.unpack);
                if (st != CUTENSOR_STATUS_SUCCESS) return st;
            }
        }
    }
    const uint64_t fixed = (uint64_t)off;
    if (fixed > devLimit) return CUTENSOR_STATUS_INSUFFICIENT_WORKSPACE;

    // the local contraction: own blocks of A and B in place, the partial (or the user's D) as packed full-size C
    const Box aBox = cell_box(d.A, cell_of_rank(d.A, me)), bBox = cell_box(d.B, cell_of_rank(d.B, me));
    const std::vector<int64_t> extA = sizes(aBox), extB = sizes(bBox);
    cutensorTensorDescriptor_t dT[3] = {nullptr, nullptr, nullptr};
    st = cutensorCreateTensorDescriptor(h, &dT[0], d.A.n, extA.data(), d.A.elemStride.data(), d.A.dtype, 256);
    if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreateTensorDescriptor(h, &dT[1], d.B.n, extB.data(), d.B.elemStride.data(), d.B.dtype, 256);
    if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreateTensorDescriptor(h, &dT[2], d.C.n, full.data(), pStride.data(), d.C.dtype, 256);
    cutensorOperationDescriptor_t op = nullptr;
    cutensorPlanPreference_t pp = nullptr;
    if (st == CUTENSOR_STATUS_SUCCESS)
        st = cutensorCreateContraction(h, &op, dT[0], d.mA.data(), d.opA, dT[1], d.mB.data(), d.opB, dT[2], d.mC.data(), d.opC,
                                       dT[2], d.mC.data(), d.compute);
    if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreatePlanPreference(h, &pp, CUTENSOR_ALGO_DEFAULT, CUTENSOR_JIT_MODE_NONE);
    uint64_t want = 0;
    if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorEstimateWorkspaceSize(h, op, pp, CUTENSOR_WORKSPACE_DEFAULT, &want);
    if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreatePlan(h, &pl->contraction, op, pp, std::min<uint64_t>(want, devLimit - fixed));
    if (st == CUTENSOR_STATUS_SUCCESS)
        st = cutensorPlanGetAttribute(h, pl->contraction, CUTENSOR_PLAN_REQUIRED_WORKSPACE, &pl->contractionWs, sizeof(uint64_t));
    cutensorDestroyOperationDescriptor(op);
    cutensorDestroyPlanPreference(pp);
    for (auto t : dT) cutensorDestroyTensorDescriptor(t);
    if (st != CUTENSOR_STATUS_SUCCESS) return st;
    pl->stageBytes = off;    // everything before the contraction workspace
    pl->compute = true;
    pl->requiredDevice = fixed + (uint64_t)round_up((int64_t)pl->contractionWs, 256);
    return CUTENSOR_STATUS_SUCCESS;
}

bool scalar_is_zero(const void* x, hipDataType t) {
    switch (t) {
        case HIP_R_64F: return *static_cast<const double*>(x) == 0.0;
        case HIP_C_64F: return static_cast<const double*>(x)[0] == 0.0 && static_cast<const double*>(x)[1] == 0.0;
        case HIP_C_32F: return static_cast<const float*>(x)[0] == 0.f && static_cast<const float*>(x)[1] == 0.f;
        default: return *static_cast<const float*>(x) == 0.f;
    }
}

cutensorStatus_t run_reduce(cutensorMpHandle* handle, const cutensorMpPlan* plan, const void* alpha, const void* A, const void* B,
                            const void* beta, const void* C, void* D, char* ws) {
    const cutensorMpOperationDescriptor& d = plan->desc;
    const ReducePlan& R = plan->red;
    cutensorHandle_t h = handle->h;
    hipStream_t s = handle->stream;
    Transport& tp = *handle->transport;
    const hipDataType dt = d.C.dtype, rt = real_type(dt);
    const int64_t es = (int64_t)elem_size(dt);
    const size_t perElem = is_complex(dt) ? 2 : 1;
    const double zero[2] = {0.0, 0.0};                  // a zero scalar of every scalar type
    // alpha / beta are of the operation's SCALAR type, which is not always the data type: fp32 data with
    // CUTENSOR_COMPUTE_DESC_64F takes double scalars (the single-GPU layer's rule, api.cpp scalar_type_for)
    const hipDataType scalarType = is_complex(dt) ? dt : ((dt == HIP_R_64F || d.compute == CUTENSOR_COMPUTE_DESC_64F) ? HIP_R_64F : HIP_R_32F);
    const bool haveBeta = !scalar_is_zero(beta, scalarType);
    char* ctrWs = ws + plan->stageBytes;
    int64_t cElems = 1;
    for (int64_t e : d.C.extent) cElems *= e;
    cutensorStatus_t st = CUTENSOR_STATUS_SUCCESS;
    if (C == nullptr) C = D;

    if (R.replicatedC) {
        char* T = R.direct ? static_cast<char*>(D) : ws + R.stageOff;
        const bool mine = haveBeta && plan->rank == 0;  // beta * C enters the sum once
        if (!R.direct && mine) { st = run_copy(h, R.cIn, dt, C, T, s); if (st != CUTENSOR_STATUS_SUCCESS) return st; }
        const void* cIn = R.direct ? C : T;
        st = cutensorContract(h, plan->contraction, alpha, A, B, mine ? beta : zero, cIn, T, ctrWs, plan->contractionWs, s);
        if (st != CUTENSOR_STATUS_SUCCESS) return st;
        if (!tp.all_reduce(T, (size_t)cElems * perElem, rt, ws + R.scratchOff, s)) return CUTENSOR_STATUS_EXECUTION_FAILED;
        if (!R.direct) st = run_copy(h, R.out, dt, T, D, s);
        return st;
    }
    char* P = ws + R.stageOff;
    const bool mine = haveBeta && R.cIn.plan != nullptr;
    if (haveBeta) {
        if (hipMemsetAsync(P, 0, (size_t)(cElems * es), s) != hipSuccess) return CUTENSOR_STATUS_EXECUTION_FAILED;
        if (mine) { st = run_copy(h, R.cIn, dt, C, P, s); if (st != CUTENSOR_STATUS_SUCCESS) return st; }
    }
    st = cutensorContract(h, plan->contraction, alpha, A, B, haveBeta ? beta : zero, P, P, ctrWs, plan->contractionWs, s);
    if (st != CUTENSOR_STATUS_SUCCESS) return st;
    const char* send = P;
    if (!R.packDirect) {
        send = ws + R.sendOff;
        for (const CopyOp& c : R.pack) {
            if (c.plan == nullptr) continue;
            st = run_copy(h, c, dt, P, ws + R.sendOff, s);
            if (st != CUTENSOR_STATUS_SUCCESS) return st;
        }
    }
    char* recv = R.recvDirect ? static_cast<char*>(D) : ws + R.recvOff;
    if (!tp.reduce_scatter(send, recv, (size_t)R.slotElems * perElem, rt, s)) return CUTENSOR_STATUS_EXECUTION_FAILED;
    if (!R.recvDirect && R.unpack.plan != nullptr) st = run_copy(h, R.unpack, dt, recv, D, s);
    return st;
}

}  // namespace

extern "C" {

static cutensorStatus_t finish_handle(cutensorMpHandle* h, int localDevice, hipStream_t stream, cutensorMpHandle_t* out) {
    h->device = localDevice;
    h->stream = stream;
    int saved = -1;
    (void)hipGetDevice(&saved);
    if (hipSetDevice(localDevice) != hipSuccess) {
        (void)hipGetLastError();
        int count = 0;
        const bool noGpu = hipGetDeviceCount(&count) != hipSuccess || count == 0;
        (void)hipGetLastError();
        // a machine without any GPU can still build and inspect plans (the planner of the single-GPU library works there
        // too); execution fails in the first HIP call.  A wrong device index on a machine with GPUs is an error.
        if (!noGpu) { delete h; return CUTENSOR_STATUS_INVALID_VALUE; }
        h->device = -1;
    }
    cutensorStatus_t st = cutensorCreate(&h->h);
    if (saved >= 0) (void)hipSetDevice(saved);
    if (st != CUTENSOR_STATUS_SUCCESS) { delete h; return st; }
    *out = h;
    return CUTENSOR_STATUS_SUCCESS;
}

// cutensorMp_contraction.cu:470-471
cutensorStatus_t cutensorMpCreate(cutensorMpHandle_t* handle, ncclComm_t comm, int localDevice, cudaStream_t stream) try {
    if (handle == nullptr || comm == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    int rank = 0, n = 0;
    if (ncclCommUserRank(comm, &rank) != ncclSuccess || ncclCommCount(comm, &n) != ncclSuccess || n <= 0)
        return CUTENSOR_STATUS_INVALID_VALUE;
    cutensorMpHandle* h = new (std::nothrow) cutensorMpHandle();
    if (h == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    h->transport = new (std::nothrow) RcclTransport(comm, rank, n);
    if (h->transport == nullptr) { delete h; return CUTENSOR_STATUS_ALLOC_FAILED; }
    return finish_handle(h, localDevice, stream, handle);
} CTAMD_API_CATCH

cutensorStatus_t cutensorMpDestroy(cutensorMpHandle_t handle) try {
    delete handle;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

cutensorStatus_t ctamdMpLocalWorldCreate(void** world, int nranks) try {
    if (world == nullptr || nranks <= 0) return CUTENSOR_STATUS_INVALID_VALUE;
    LocalWorld* w = new (std::nothrow) LocalWorld(nranks);
    if (w == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    *world = w;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH
cutensorStatus_t ctamdMpLocalWorldDestroy(void* world) try {
    delete static_cast<LocalWorld*>(world);
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH
cutensorStatus_t ctamdMpCreateOnLocalWorld(cutensorMpHandle_t* handle, void* world, int rank, int localDevice, cudaStream_t stream) try {
    LocalWorld* w = static_cast<LocalWorld*>(world);
    if (handle == nullptr || w == nullptr || rank < 0 || rank >= w->nranks) return CUTENSOR_STATUS_INVALID_VALUE;
    cutensorMpHandle* h = new (std::nothrow) cutensorMpHandle();
    if (h == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    LocalTransport* lt = new (std::nothrow) LocalTransport(w, rank);
    h->transport = lt;
    if (h->transport == nullptr) { delete h; return CUTENSOR_STATUS_ALLOC_FAILED; }
    cutensorStatus_t st = finish_handle(h, localDevice, stream, handle);
    if (st == CUTENSOR_STATUS_SUCCESS) lt->set_engine((*handle)->h);
    return st;
} CTAMD_API_CATCH

// cutensorMp_contraction.cu:474-483
cutensorStatus_t cutensorMpCreateTensorDescriptor(const cutensorMpHandle_t handle, cutensorMpTensorDescriptor_t* desc,
                                                  uint32_t numModes, const int64_t extent[], const int64_t elementStride[],
                                                  const int64_t blockSize[], const int64_t blockStride[],
                                                  const int64_t nranksPerMode[], uint32_t nranks, const int32_t ranks[],
                                                  cutensorDataType_t type) try {
    (void)blockStride;
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || (numModes > 0 && extent == nullptr)) return CUTENSOR_STATUS_INVALID_VALUE;
    if (numModes > 62 || elem_size(type) == 0) return CUTENSOR_STATUS_NOT_SUPPORTED;
    cutensorMpTensorDescriptor* d = new (std::nothrow) cutensorMpTensorDescriptor();
    if (d == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    MpTensor& t = d->t;
    t.n = numModes;
    t.dtype = type;
    t.extent.assign(extent, extent + numModes);
    t.p.resize(numModes); t.bs.resize(numModes); t.elemStride.resize(numModes); t.cellStride.resize(numModes);
    int64_t cells = 1, run = 1;
    for (uint32_t i = 0; i < numModes; ++i) {
        t.p[i] = nranksPerMode ? nranksPerMode[i] : 1;
        if (extent[i] <= 0 || t.p[i] <= 0) { delete d; return CUTENSOR_STATUS_INVALID_VALUE; }
        t.bs[i] = blockSize ? blockSize[i] : (extent[i] + t.p[i] - 1) / t.p[i];
        if (t.bs[i] <= 0) { delete d; return CUTENSOR_STATUS_INVALID_VALUE; }
        if (t.bs[i] * t.p[i] < extent[i]) { delete d; return CUTENSOR_STATUS_NOT_SUPPORTED; }   // block-cyclic
        t.cellStride[i] = cells;
        cells *= t.p[i];
        t.elemStride[i] = elementStride ? elementStride[i] : run;
        if (t.elemStride[i] != run) t.packed = false;
        run *= t.bs[i];
    }
    t.numCells = cells;
    const int world = handle->transport->size();
    if (cells > 1) {
        if ((int64_t)nranks != cells || cells != world) { delete d; return CUTENSOR_STATUS_INVALID_VALUE; }
        t.owner.resize((size_t)cells);
        std::vector<char> seen((size_t)world, 0);
        for (int64_t c = 0; c < cells; ++c) {
            const int32_t r = ranks ? ranks[c] : (int32_t)c;
            if (r < 0 || r >= world || seen[(size_t)r]) { delete d; return CUTENSOR_STATUS_INVALID_VALUE; }   // one block per rank
            seen[(size_t)r] = 1;
            t.owner[(size_t)c] = r;
        }
    }
    *desc = d;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

cutensorStatus_t cutensorMpDestroyTensorDescriptor(cutensorMpTensorDescriptor_t desc) try {
    delete desc;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// cutensorMp_contraction.cu:485-488
cutensorStatus_t cutensorMpCreateContraction(const cutensorMpHandle_t handle, cutensorMpOperationDescriptor_t* desc,
                                             const cutensorMpTensorDescriptor_t descA, const int32_t modesA[], cutensorOperator_t opA,
                                             const cutensorMpTensorDescriptor_t descB, const int32_t modesB[], cutensorOperator_t opB,
                                             const cutensorMpTensorDescriptor_t descC, const int32_t modesC[], cutensorOperator_t opC,
                                             const cutensorMpTensorDescriptor_t descD, const int32_t modesD[],
                                             const cutensorComputeDescriptor_t compute) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (desc == nullptr || descA == nullptr || descB == nullptr || descC == nullptr || descD == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    if ((descA->t.n && !modesA) || (descB->t.n && !modesB) || (descC->t.n && (!modesC || !modesD))) return CUTENSOR_STATUS_INVALID_VALUE;
    cutensorMpOperationDescriptor* d = new (std::nothrow) cutensorMpOperationDescriptor();
    if (d == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    d->A = descA->t; d->B = descB->t; d->C = descC->t;
    d->mA.assign(modesA, modesA + d->A.n);
    d->mB.assign(modesB, modesB + d->B.n);
    d->mC.assign(modesC, modesC + d->C.n);
    d->opA = opA; d->opB = opB; d->opC = opC;
    d->compute = compute;
    const MpTensor& D = descD->t;
    const std::vector<int32_t> mD(modesD, modesD + D.n);
    const bool sameCD = mD == d->mC && D.extent == d->C.extent && D.p == d->C.p && D.bs == d->C.bs &&
                        D.elemStride == d->C.elemStride && D.owner == d->C.owner && D.dtype == d->C.dtype;
    if (!sameCD || d->A.dtype != d->B.dtype || d->A.dtype != d->C.dtype) { delete d; return CUTENSOR_STATUS_NOT_SUPPORTED; }
    auto check = [&](const MpTensor& x, const std::vector<int32_t>& mx, const MpTensor& y, const std::vector<int32_t>& my) {
        for (uint32_t i = 0; i < x.n; ++i) {
            const int j = find_label(my, mx[i]);
            if (j >= 0 && x.extent[i] != y.extent[(size_t)j]) return false;
        }
        return true;
    };
    if (!check(d->A, d->mA, d->B, d->mB) || !check(d->A, d->mA, d->C, d->mC) || !check(d->B, d->mB, d->C, d->mC)) {
        delete d;
        return CUTENSOR_STATUS_INVALID_VALUE;
    }
    for (int32_t l : d->mC)
        if (find_label(d->mA, l) < 0 && find_label(d->mB, l) < 0) { delete d; return CUTENSOR_STATUS_INVALID_VALUE; }
    *desc = d;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

cutensorStatus_t cutensorMpDestroyOperationDescriptor(cutensorMpOperationDescriptor_t desc) try {
    delete desc;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// cutensorMp_contraction.cu:490-500
cutensorStatus_t cutensorMpCreatePlanPreference(const cutensorMpHandle_t handle, cutensorMpPlanPreference_t* pref,
                                                cutensorMpAlgo_t algo, uint64_t workspaceSizeDeviceLimit,
                                                uint64_t workspaceSizeHostLimit) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (pref == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    if (algo != CUTENSORMP_ALGO_DEFAULT) return CUTENSOR_STATUS_NOT_SUPPORTED;
    cutensorMpPlanPreference* p = new (std::nothrow) cutensorMpPlanPreference{algo, workspaceSizeDeviceLimit, workspaceSizeHostLimit};
    if (p == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    *pref = p;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

cutensorStatus_t cutensorMpDestroyPlanPreference(cutensorMpPlanPreference_t pref) try {
    delete pref;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// cutensorMp_contraction.cu:502-503
cutensorStatus_t cutensorMpCreatePlan(const cutensorMpHandle_t handle, cutensorMpPlan_t* plan,
                                      const cutensorMpOperationDescriptor_t desc, const cutensorMpPlanPreference_t pref) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (plan == nullptr || desc == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    const uint64_t devLimit = pref ? pref->devLimit : (1ull << 62);
    int saved = -1;
    (void)hipGetDevice(&saved);
    if (handle->device >= 0) (void)hipSetDevice(handle->device);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{saved};

    cutensorMpPlan* pl = new (std::nothrow) cutensorMpPlan();
    if (pl == nullptr) return CUTENSOR_STATUS_ALLOC_FAILED;
    pl->desc = *desc;
    const cutensorMpOperationDescriptor& d = pl->desc;
    const int me = pl->rank = handle->transport->rank();
    const int world = pl->nranks = handle->transport->size();
    const int64_t es = (int64_t)elem_size(d.A.dtype);
    const MpTensor* X[2] = {&d.A, &d.B};
    cutensorHandle_t h = handle->h;
    cutensorStatus_t st = CUTENSOR_STATUS_SUCCESS;
    auto bail = [&](cutensorStatus_t s) { delete pl; return s; };

    // ---- the box of C every rank owns, hence the boxes of A and B it needs ---------------------------------
    std::vector<Box> cBox((size_t)world);
    for (int q = 0; q < world; ++q) {
        const int64_t c = cell_of_rank(d.C, q);
        if (c < 0) return bail(CUTENSOR_STATUS_INTERNAL_ERROR);
        cBox[(size_t)q] = cell_box(d.C, c);
    }
    pl->compute = !cBox[(size_t)me].empty();

    // ---- algorithm: identical inputs on every rank, hence an identical choice -------------------------------------
    {
        int64_t cElems = 1;
        for (int64_t e : d.C.extent) cElems *= e;
        pl->gatherTotal = gather_traffic(d, cBox, world, es);
        // ring all-reduce moves ~2 |C| per rank, reduce-scatter ~|C|
        pl->reduceTotal = (int64_t)world * cElems * es * (d.C.replicated() ? 2 : 1);
        bool useReduce = reduce_applicable(d, world) && pl->reduceTotal < pl->gatherTotal;
        if (const char* force = CTAMD_HOOK_ENV("CUTENSORMP_AMD_ALGO")) {       // tests: "gather" / "reduce" on every rank
            if (std::strcmp(force, "gather") == 0) useReduce = false;
            else if (std::strcmp(force, "reduce") == 0) useReduce = reduce_applicable(d, world);
        }
        if (useReduce) {
            pl->reduce = true;
            st = build_reduce_plan(handle, pl, cBox, devLimit);
            if (st != CUTENSOR_STATUS_SUCCESS) return bail(st);
            *plan = pl;
            return CUTENSOR_STATUS_SUCCESS;
        }
    }

    // ---- transfers: (tensor, receiver q, cell c) in a fixed order both sides agree on ------------------------
    int64_t sendOff = 0, recvOff = 0;
    for (int k = 0; k < 2; ++k) {
        const MpTensor& x = *X[k];
        const Box myCell = cell_box(x, cell_of_rank(x, me));
        const std::vector<int64_t> userStride = x.elemStride;
        for (int q = 0; q < world; ++q) {
            if (cBox[(size_t)q].empty()) continue;
            const Box need = needed_box(d, k, cBox[(size_t)q]);
            if (q == me) {
                OperandPlan& o = pl->in[k];
                o.need = need;
                const Box own = intersect(need, myCell);
                o.staged = !(own == need);        // in place only when the rank's own block covers the whole box
                if (!o.staged) { o.viewOff = box_offset(need, myCell, userStride); o.viewStride = userStride; }
            }
            if (x.replicated()) continue;        // every rank already holds all of it
            for (int64_t c = 0; c < x.numCells; ++c) {
                const int src = x.owner[(size_t)c];
                if (src == q || (src != me && q != me)) continue;
                const Box cb = cell_box(x, c);
                const Box part = intersect(need, cb);
                if (part.empty()) continue;
                Transfer t;
                t.tensor = k; t.src = src; t.dst = q; t.box = part;
                t.bytes = part.volume() * es;
                if (src == me) {
                    t.direct = x.packed && part == cb && sizes(cb) == x.bs;
                    if (!t.direct) {
                        t.sendOff = sendOff; sendOff += round_up(t.bytes, 256);
                        st = make_copy(h, x.dtype, copy_modes(part, userStride, packed_strides(sizes(part))), t.pack,
                                       box_offset(part, cb, userStride), 0);
                        if (st != CUTENSOR_STATUS_SUCCESS) return bail(st);
                    }
                    pl->sends.push_back(std::move(t));
                } else {
                    t.inPlace = part == need;
                    if (!t.inPlace) { t.recvOff = recvOff; recvOff += round_up(t.bytes, 256); }
                    pl->recvs.push_back(std::move(t));
                }
            }
        }
    }
    pl->sendBytes = sendOff;
    pl->recvBytes = recvOff;

    // ---- staging tensors, local copies, unpack copies ---------------------------------------------------------
    int64_t stageOff = 0;
    for (int k = 0; k < 2 && pl->compute; ++k) {
        const MpTensor& x = *X[k];
        OperandPlan& o = pl->in[k];
        if (!o.staged) continue;
        const std::vector<int64_t> stStride = packed_strides(sizes(o.need));
        o.stageOff = stageOff;
        o.stageBytes = round_up(o.need.volume() * es, 256);
        stageOff += o.stageBytes;
        o.viewStride = stStride;
        const Box myCell = cell_box(x, cell_of_rank(x, me));
        const Box own = intersect(o.need, myCell);
        if (!own.empty()) {
            CopyOp c;
            st = make_copy(h, x.dtype, copy_modes(own, x.elemStride, stStride), c, box_offset(own, myCell, x.elemStride),
                           box_offset(own, o.need, stStride));
            if (st != CUTENSOR_STATUS_SUCCESS) return bail(st);
            o.localCopies.push_back(std::move(c));
        }
        for (Transfer& t : pl->recvs) {
            if (t.tensor != k || t.inPlace) continue;
            st = make_copy(h, x.dtype, copy_modes(t.box, packed_strides(sizes(t.box)), stStride), t.unpack, 0,
                           box_offset(t.box, o.need, stStride));
            if (st != CUTENSOR_STATUS_SUCCESS) return bail(st);
        }
    }
    pl->stageBytes = stageOff;
    const uint64_t fixed = (uint64_t)(pl->sendBytes + pl->recvBytes + pl->stageBytes);
    if (fixed > devLimit) return bail(CUTENSOR_STATUS_INSUFFICIENT_WORKSPACE);

    // ---- the local contraction -----------------------------------------------------------------------------------
    if (pl->compute) {
        const Box& cb = cBox[(size_t)me];
        cutensorTensorDescriptor_t dT[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < 2 && st == CUTENSOR_STATUS_SUCCESS; ++k) {
            const OperandPlan& o = pl->in[k];
            const std::vector<int64_t> ext = sizes(o.need);
            const uint32_t align = o.staged ? 256u : alignment_of(o.viewOff * es);
            st = cutensorCreateTensorDescriptor(h, &dT[k], X[k]->n, ext.data(), o.viewStride.data(), X[k]->dtype, align);
        }
        const std::vector<int64_t> extC = sizes(cb);
        if (st == CUTENSOR_STATUS_SUCCESS)
            st = cutensorCreateTensorDescriptor(h, &dT[2], d.C.n, extC.data(), d.C.elemStride.data(), d.C.dtype, 256);
        cutensorOperationDescriptor_t op = nullptr;
        cutensorPlanPreference_t pp = nullptr;
        if (st == CUTENSOR_STATUS_SUCCESS)
            st = cutensorCreateContraction(h, &op, dT[0], d.mA.data(), d.opA, dT[1], d.mB.data(), d.opB, dT[2], d.mC.data(), d.opC,
                                           dT[2], d.mC.data(), d.compute);
        if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorCreatePlanPreference(h, &pp, CUTENSOR_ALGO_DEFAULT, CUTENSOR_JIT_MODE_NONE);
        uint64_t want = 0;
        if (st == CUTENSOR_STATUS_SUCCESS) st = cutensorEstimateWorkspaceSize(h, op, pp, CUTENSOR_WORKSPACE_DEFAULT, &want);
        if (st == CUTENSOR_STATUS_SUCCESS) {
            const uint64_t limit = std::min<uint64_t>(want, devLimit - fixed);
            st = cutensorCreatePlan(h, &pl->contraction, op, pp, limit);
        }
        if (st == CUTENSOR_STATUS_SUCCESS)
            st = cutensorPlanGetAttribute(h, pl->contraction, CUTENSOR_PLAN_REQUIRED_WORKSPACE, &pl->contractionWs, sizeof(uint64_t));
        cutensorDestroyOperationDescriptor(op);
        cutensorDestroyPlanPreference(pp);
        for (auto t : dT) cutensorDestroyTensorDescriptor(t);
        if (st != CUTENSOR_STATUS_SUCCESS) return bail(st);
    }
    pl->requiredDevice = fixed + (uint64_t)round_up((int64_t)pl->contractionWs, 256);
    *plan = pl;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// cutensorMp_contraction.cu:506-509
cutensorStatus_t cutensorMpPlanGetAttribute(const cutensorMpHandle_t handle, const cutensorMpPlan_t plan,
                                            cutensorMpPlanAttribute_t attr, void* buf, size_t sizeInBytes) try {
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (plan == nullptr || buf == nullptr || sizeInBytes < sizeof(uint64_t)) return CUTENSOR_STATUS_INVALID_VALUE;
    switch (attr) {
        case CUTENSORMP_PLAN_REQUIRED_WORKSPACE_DEVICE: *static_cast<uint64_t*>(buf) = plan->requiredDevice; return CUTENSOR_STATUS_SUCCESS;
        case CUTENSORMP_PLAN_REQUIRED_WORKSPACE_HOST: *static_cast<uint64_t*>(buf) = 0; return CUTENSOR_STATUS_SUCCESS;
    }
    return CUTENSOR_STATUS_INVALID_VALUE;
} CTAMD_API_CATCH

cutensorStatus_t cutensorMpDestroyPlan(cutensorMpPlan_t plan) try {
    delete plan;
    return CUTENSOR_STATUS_SUCCESS;
} CTAMD_API_CATCH

// cutensorMp_contraction.cu:537-538
cutensorStatus_t cutensorMpContract(const cutensorMpHandle_t handle, const cutensorMpPlan_t plan, const void* alpha,
                                    const void* A, const void* B, const void* beta, const void* C, void* D,
                                    void* workspaceDevice, void* workspaceHost) try {
    (void)workspaceHost;
    if (handle == nullptr) return CUTENSOR_STATUS_NOT_INITIALIZED;
    if (plan == nullptr || alpha == nullptr || beta == nullptr || A == nullptr || B == nullptr || D == nullptr) return CUTENSOR_STATUS_INVALID_VALUE;
    if (plan->requiredDevice > 0 && workspaceDevice == nullptr) return CUTENSOR_STATUS_INSUFFICIENT_WORKSPACE;
    if (handle->device < 0) return CUTENSOR_STATUS_ARCH_MISMATCH;   // plan-only handle of a machine without a GPU
    int saved = -1;
    (void)hipGetDevice(&saved);
    if (handle->device >= 0) (void)hipSetDevice(handle->device);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{saved};

    const cutensorMpOperationDescriptor& d = plan->desc;
    const int64_t es = (int64_t)elem_size(d.A.dtype);
    hipStream_t s = handle->stream;
    cutensorHandle_t h = handle->h;
    Transport& tp = *handle->transport;
    char* ws = static_cast<char*>(workspaceDevice);
    char* sendBase = ws;
    char* recvBase = sendBase + plan->sendBytes;
    char* stageBase = recvBase + plan->recvBytes;
    char* ctrWs = stageBase + plan->stageBytes;
    const void* user[2] = {A, B};
    cutensorStatus_t st = CUTENSOR_STATUS_SUCCESS;
    if (plan->reduce) return run_reduce(handle, plan, alpha, A, B, beta, C, D, ws);

    // 1. pack what other ranks need from this rank's blocks
    for (const Transfer& t : plan->sends) {
        if (t.direct) continue;
        st = run_copy(h, t.pack, d.A.dtype, user[t.tensor], sendBase + t.sendOff, s);
        if (st != CUTENSOR_STATUS_SUCCESS) return st;
    }
    // 2. one group of point-to-point transfers
    if (!plan->sends.empty() || !plan->recvs.empty()) {
        if (!tp.begin()) return CUTENSOR_STATUS_EXECUTION_FAILED;
        bool ok = true;
        for (const Transfer& t : plan->sends)
            ok = ok && tp.send(t.direct ? user[t.tensor] : sendBase + t.sendOff, (size_t)t.bytes, t.dst, s);
        for (const Transfer& t : plan->recvs) {
            char* dst = t.inPlace ? stageBase + plan->in[t.tensor].stageOff : recvBase + t.recvOff;
            ok = ok && tp.recv(dst, (size_t)t.bytes, t.src, s);
        }
        ok = tp.end(s) && ok;
        if (!ok) return CUTENSOR_STATUS_EXECUTION_FAILED;
    }
    if (!plan->compute) return CUTENSOR_STATUS_SUCCESS;
    // 3. assemble the staged operands
    for (int k = 0; k < 2; ++k) {
        const OperandPlan& o = plan->in[k];
        if (!o.staged) continue;
        char* stage = stageBase + o.stageOff;
        for (const CopyOp& c : o.localCopies) {
            st = run_copy(h, c, d.A.dtype, user[k], stage, s);
            if (st != CUTENSOR_STATUS_SUCCESS) return st;
        }
        for (const Transfer& t : plan->recvs) {
            if (t.tensor != k || t.inPlace) continue;
            st = run_copy(h, t.unpack, d.A.dtype, recvBase + t.recvOff, stage, s);
            if (st != CUTENSOR_STATUS_SUCCESS) return st;
        }
    }
    // 4. the local contraction writes this rank's block of D
    const void* opnd[2];
    for (int k = 0; k < 2; ++k) {
        const OperandPlan& o = plan->in[k];
        opnd[k] = o.staged ? static_cast<const void*>(stageBase + o.stageOff)
                           : static_cast<const void*>(static_cast<const char*>(user[k]) + o.viewOff * es);
    }
    return cutensorContract(h, plan->contraction, alpha, opnd[0], opnd[1], beta, C ? C : D, D, ctrWs, plan->contractionWs, s);
} CTAMD_API_CATCH

size_t ctamdMpDescribePlan(const cutensorMpPlan_t plan, char* buf, size_t bufSize) try {
    if (plan == nullptr) return 0;
    std::string j = "{";
    char tmp[256];
    auto add = [&](const char* fmt, auto... a) { std::snprintf(tmp, sizeof(tmp), fmt, a...); j += tmp; };
    add("\"rank\": %d, \"nranks\": %d, \"compute\": %s, ", plan->rank, plan->nranks, plan->compute ? "true" : "false");
    add("\"algorithm\": \"%s\", \"gatherTotal\": %lld, \"reduceTotal\": %lld, ", plan->reduce ? "reduce" : "gather",
        (long long)plan->gatherTotal, (long long)plan->reduceTotal);
    add("\"reduceDirect\": %s, \"packDirect\": %s, \"recvDirect\": %s, ", plan->red.direct ? "true" : "false",
        plan->red.packDirect ? "true" : "false", plan->red.recvDirect ? "true" : "false");
    add("\"stagedA\": %s, \"stagedB\": %s, ", plan->in[0].staged ? "true" : "false", plan->in[1].staged ? "true" : "false");
    add("\"sendBytes\": %lld, \"recvBytes\": %lld, \"stageBytes\": %lld, \"contractionWorkspace\": %llu, \"requiredDevice\": %llu, ",
        (long long)plan->sendBytes, (long long)plan->recvBytes, (long long)plan->stageBytes,
        (unsigned long long)plan->contractionWs, (unsigned long long)plan->requiredDevice);
    auto list = [&](const char* name, const std::vector<Transfer>& v) {
        j += std::string("\"") + name + "\": [";
        for (size_t i = 0; i < v.size(); ++i) {
            const Transfer& t = v[i];
            add("%s{\"tensor\": \"%c\", \"src\": %d, \"dst\": %d, \"bytes\": %lld, \"direct\": %s, \"inPlace\": %s, \"launches\": %d}",
                i ? ", " : "", t.tensor == 0 ? 'A' : 'B', t.src, t.dst, (long long)t.bytes, t.direct ? "true" : "false",
                t.inPlace ? "true" : "false", (int)(t.pack.plan ? t.pack.launches.size() : t.unpack.plan ? t.unpack.launches.size() : 0));
        }
        j += "]";
    };
    list("sends", plan->sends);
    j += ", ";
    list("recvs", plan->recvs);
    j += "}";
    if (buf != nullptr && bufSize > 0) {
        const size_t n = std::min(bufSize - 1, j.size());
        std::memcpy(buf, j.data(), n);
        buf[n] = 0;
    }
    return j.size() + 1;
} CTAMD_API_CATCH_ZERO

}  // extern "C"
