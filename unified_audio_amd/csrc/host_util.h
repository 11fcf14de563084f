// host_util.h - host-side plumbing shared by the model graphs: weight-table lookup, the folded-weight store,
// the per-call bump arena and the planning / execution context.
#pragma once
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.h"

namespace qa {

// name -> host tensor view of the caller's state_dict
class HostTable {
   public:
    HostTable(const qa_tensor* t, int64_t n) {
        for (int64_t i = 0; i < n; ++i)
            if (t[i].name && t[i].data) map_[t[i].name] = &t[i];
    }
    // nullptr + error message when absent or mis-sized
    const float* get(const std::string& name, int64_t numel) const {
        auto it = map_.find(name);
        if (it == map_.end()) {
            set_error("weight table: missing tensor '%s'", name.c_str());
            return nullptr;
        }
        if (it->second->numel != numel) {
            set_error("weight table: tensor '%s' has %lld elements, expected %lld", name.c_str(),
                      (long long)it->second->numel, (long long)numel);
            return nullptr;
        }
        return it->second->data;
    }
    bool has(const std::string& name) const { return map_.count(name) != 0; }

   private:
    std::unordered_map<std::string, const qa_tensor*> map_;
};

// Folded weights are appended to one host blob (64-float aligned) and uploaded with a single copy.
class WeightStore {
   public:
    size_t add(const std::vector<float>& v) { return add(v.data(), v.size()); }
    size_t add(const float* p, size_t n) {
        const size_t off = blob_.size();
        blob_.insert(blob_.end(), p, p + n);
        blob_.resize(round_up((int64_t)blob_.size(), 64), 0.f);
        return off;
    }
    int upload() {
        QA_HIP(hipMalloc(&dev_, blob_.size() * sizeof(float)));
        QA_HIP(hipMemcpy(dev_, blob_.data(), blob_.size() * sizeof(float), hipMemcpyHostToDevice));
        bytes_ = blob_.size() * sizeof(float);
        std::vector<float>().swap(blob_);
        return QA_OK;
    }
    const float* ptr(size_t off) const { return dev_ + off; }
    size_t bytes() const { return bytes_; }
    void release() {
        if (dev_) (void)hipFree(dev_);
        dev_ = nullptr;
    }

   private:
    std::vector<float> blob_;
    float* dev_ = nullptr;
    size_t bytes_ = 0;
};

// Bump allocator over one device buffer.  In planning mode (base == nullptr) it only tracks the peak.
class Arena {
   public:
    void begin(char* base, size_t cap) {
        base_ = base;
        cap_ = cap;
        off_ = 0;
        peak_ = 0;
        floor_ = 0;
    }
    template <typename T>
    T* alloc(size_t n) {
        const size_t bytes = (size_t)round_up((int64_t)(n * sizeof(T)), 256);
        const size_t at = off_;
        off_ += bytes;
        if (off_ > peak_) peak_ = off_;
        return base_ ? reinterpret_cast<T*>(base_ + at) : reinterpret_cast<T*>(uintptr_t(4096) + at);
    }
    size_t mark() const { return off_; }
    void release(size_t m) { off_ = m > floor_ ? m : floor_; }
    // everything allocated so far survives later release() calls (test taps snapshot buffers inside a mark / release region)
    void pin() { floor_ = off_; }
    size_t peak() const { return peak_; }
    bool planning() const { return base_ == nullptr; }

   private:
    char* base_ = nullptr;
    size_t cap_ = 0, off_ = 0, peak_ = 0, floor_ = 0;
};

struct Tap {
    const float* ptr;
    int64_t numel;
};

struct Ctx {
    Arena arena;
    hipStream_t stream = nullptr;
    bool dry = false;      // planning pass: allocate, do not launch
    bool capture = false;  // test hook: snapshot named intermediates (buffers are reused / updated in place later)
    std::unordered_map<std::string, Tap> taps;
    void tap(const std::string& name, const float* p, int64_t n) {
        if (!capture) return;
        float* copy = arena.alloc<float>((size_t)n);
        arena.pin();  // the snapshot must outlive the mark / release region it was taken in
        if (dry) return;
        (void)hipMemcpyAsync(copy, p, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, stream);
        taps[name] = Tap{copy, n};
    }
};

// A folded convolution / linear layer in library layout [N][ksize][C_in].
struct ConvW {
    const float* w = nullptr;
    const float* b = nullptr;
    int N = 0, C_in = 0, ksize = 1;
    int algo_n = 0, algo_cin = 0;  // un-padded sizes (0 = same as N / C_in)
};

inline int pad32(int c) { return (int)round_up(c, 32); }

}  // namespace qa
