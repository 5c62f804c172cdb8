// host_xfer.hip -- the host-pointer boundary's transfers (rsdsfm_ransac / rsdsfm_refine / rsdsfm_estimate_inverse_depths ...: what
// minimal::ransac and nonLinearRefinement in host/*.h call, reference main.cc:447-457).
//
// The reference's API hands over pageable host arrays (Eigen objects) and wants pageable arrays back; a 1280x720 pair moves ~44 MB in and
// ~67 MB out.  hipMemcpyAsync on pageable memory stages through the runtime's own bounce buffer on ONE thread (tools/xfer_probe.hip: the rate it
// reaches on the box is in profiles/r06_xfer_probe.txt), well below PCIe.  Here: a ring of pinned chunks per process, filled / drained by a small
// pool of host threads while the DMA engine moves the neighbouring chunk, on the context's stream so that kernels queue behind the last chunk
// with no extra synchronisation.  h2d returns once the caller's array has been READ (the DMA may still be in flight); d2h returns once the
// caller's array has been WRITTEN.  Small copies (< 256 KB) take the plain path.
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "rsdsfm_internal.hpp"

namespace rsdsfm {

namespace {

constexpr size_t kXferChunk = (size_t)4 << 20;  // bytes per pinned chunk
constexpr int kXferSlots = 4;                   // chunks in the ring
constexpr size_t kXferSmall = (size_t)256 << 10;

// a few persistent workers that copy pieces of one chunk (a fork / join per chunk: ~10 us)
class CopyPool {
public:
    explicit CopyPool(int n) : stop_(false), pending_(0), epoch_(0) {
        for (int i = 0; i < n; ++i) workers_.emplace_back([this, i, n] { run(i, n); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    int size() const { return (int)workers_.size(); }
    // dst[0 .. n) = src[0 .. n), split over the workers and the calling thread
    void copy(char* dst, const char* src, size_t n) {
        const int parts = size() + 1;
        if (n < ((size_t)64 << 10) || parts == 1) {
            memcpy(dst, src, n);
            return;
        }
        const size_t per = ((n + parts - 1) / parts + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> g(m_);
            dst_ = dst, src_ = src, n_ = n, per_ = per;
            pending_ = size();
            ++epoch_;
        }
        cv_.notify_all();
        const size_t a = std::min(n, per * (size_t)size());
        if (n > a) memcpy(dst + a, src + a, n - a);  // (the caller takes the last piece)
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
    }

private:
    void run(int i, int n) {
        (void)n;
        unsigned long long seen = 0;
        for (;;) {
            char* d;
            const char* s;
            size_t len, per;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || epoch_ != seen; });
                if (stop_) return;
                seen = epoch_;
                d = dst_, s = src_, len = n_, per = per_;
            }
            const size_t a = std::min(len, per * (size_t)i), b = std::min(len, per * (size_t)(i + 1));
            if (b > a) memcpy(d + a, s + a, b - a);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    bool stop_;
    int pending_;
    unsigned long long epoch_;
    char* dst_ = nullptr;
    const char* src_ = nullptr;
    size_t n_ = 0, per_ = 0;
};

// one ring per process and device (transfers of different contexts on a device serialise on it: they share the link anyway)
struct XferRing {
    std::mutex m;
    char* pinned[kXferSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[kXferSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool busy[kXferSlots] = {false, false, false, false};
    bool ok = false, tried = false;
    CopyPool* pool = nullptr;
};
XferRing g_ring[64];

int xfer_threads() {
    static int n = -1;
    if (n < 0) {
        const char* e = getenv("RSDSFM_XFER_THREADS");  // helper threads beside the caller's (0: the caller's thread alone)
        const unsigned hw = std::thread::hardware_concurrency();
        n = e ? std::max(0, std::min(atoi(e), 32)) : (int)std::max(0u, std::min(7u, hw > 2 ? hw / 2 - 1 : 0u));
    }
    return n;
}

XferRing* ring_of(Ctx* c) {
    XferRing* R = &g_ring[c->device & 63];
    if (!R->tried) {
        R->tried = true;
        bool good = true;
        for (int s = 0; s < kXferSlots && good; ++s)
            good = hipHostMalloc((void**)&R->pinned[s], kXferChunk, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&R->ev[s], hipEventDisableTiming) == hipSuccess;
        if (good) R->pool = new (std::nothrow) CopyPool(xfer_threads());
        R->ok = good && R->pool;
    }
    return R->ok ? R : nullptr;
}

}  // namespace

// host -> device on the context's stream; returns once `h_src` has been read
int xfer_h2d(Ctx* c, void* d_dst, const void* h_src, size_t bytes) {
    if (bytes == 0) return RSDSFM_OK;
    XferRing* R = bytes >= kXferSmall ? ring_of(c) : nullptr;
    if (!R) {
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
        return RSDSFM_OK;
    }
    std::lock_guard<std::mutex> g(R->m);
    int slot = 0;
    for (size_t off = 0; off < bytes; off += kXferChunk, slot = (slot + 1) % kXferSlots) {
        const size_t len = std::min(kXferChunk, bytes - off);
        if (R->busy[slot]) RSDSFM_HIP_CHECK(c, hipEventSynchronize(R->ev[slot]));  // (the DMA that last read this chunk)
        R->pool->copy(R->pinned[slot], static_cast<const char*>(h_src) + off, len);
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(static_cast<char*>(d_dst) + off, R->pinned[slot], len, hipMemcpyHostToDevice, c->stream));
        RSDSFM_HIP_CHECK(c, hipEventRecord(R->ev[slot], c->stream));
        R->busy[slot] = true;
    }
    return RSDSFM_OK;
}

// device -> host behind everything enqueued on the context's stream; returns once `h_dst` holds the data
int xfer_d2h(Ctx* c, void* h_dst, const void* d_src, size_t bytes) {
    if (bytes == 0) return RSDSFM_OK;
    XferRing* R = bytes >= kXferSmall ? ring_of(c) : nullptr;
    if (!R) {
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
        RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        return RSDSFM_OK;
    }
    std::lock_guard<std::mutex> g(R->m);
    for (int s = 0; s < kXferSlots; ++s)  // (chunks an earlier upload still reads)
        if (R->busy[s]) {
            RSDSFM_HIP_CHECK(c, hipEventSynchronize(R->ev[s]));
            R->busy[s] = false;
        }
    const size_t nchunks = (bytes + kXferChunk - 1) / kXferChunk;
    // chunk k's DMA is enqueued kXferSlots - 1 chunks ahead of its drain: the pool copies chunk k to the caller while the engine fills the next ones
    auto issue = [&](size_t k) -> int {
        const int slot = (int)(k % kXferSlots);
        const size_t off = k * kXferChunk, len = std::min(kXferChunk, bytes - off);
        RSDSFM_HIP_CHECK(c, hipMemcpyAsync(R->pinned[slot], static_cast<const char*>(d_src) + off, len, hipMemcpyDeviceToHost, c->stream));
        RSDSFM_HIP_CHECK(c, hipEventRecord(R->ev[slot], c->stream));
        return RSDSFM_OK;
    };
    size_t issued = 0;
    for (; issued < nchunks && issued < (size_t)(kXferSlots - 1); ++issued) {
        int rc = issue(issued);
        if (rc != RSDSFM_OK) return rc;
    }
    for (size_t k = 0; k < nchunks; ++k) {
        if (issued < nchunks) {
            int rc = issue(issued++);
            if (rc != RSDSFM_OK) return rc;
        }
        const int slot = (int)(k % kXferSlots);
        const size_t off = k * kXferChunk, len = std::min(kXferChunk, bytes - off);
        RSDSFM_HIP_CHECK(c, hipEventSynchronize(R->ev[slot]));
        R->pool->copy(static_cast<char*>(h_dst) + off, R->pinned[slot], len);
    }
    return RSDSFM_OK;
}

// several device -> host copies as ONE pipeline (the drain of one array overlaps the DMA of the next)
int xfer_d2h_many(Ctx* c, const XferItem* items, int count) {
    // (simple form: back to back -- each call's first DMA is enqueued while nothing drains, a bubble of one chunk per array)
    for (int i = 0; i < count; ++i) {
        int rc = xfer_d2h(c, items[i].host, items[i].dev, items[i].bytes);
        if (rc != RSDSFM_OK) return rc;
    }
    return RSDSFM_OK;
}

}  // namespace rsdsfm
