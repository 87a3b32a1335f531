// Host-side builder of one device blob holding all packed tensors of a model.
#pragma once
#include "common.h"

namespace dgdm {

struct Blob {
    std::vector<float> host;
    DevBuf dev;
    // append `v`, 256-byte aligned; returns the offset (in floats)
    size_t add(const std::vector<float> &v) { return add(v.data(), v.size()); }
    size_t add(const float *p, size_t n) {
        size_t off = (host.size() + 63) & ~size_t(63);
        host.resize(off + n);
        for (size_t i = 0; i < n; ++i) host[off + i] = p[i];
        return off;
    }
    int upload() { host.resize((host.size() + 63) & ~size_t(63)); return dev.upload(host.data(), host.size() * sizeof(float)); }
    const float *at(size_t off) const { return dev.as<float>() + off; }
    const float4 *at4(size_t off) const { return reinterpret_cast<const float4 *>(dev.as<float>() + off); }
};

}  // namespace dgdm
