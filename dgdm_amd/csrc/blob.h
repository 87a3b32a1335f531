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

// The float64 weights of the stages that run in double precision (see Folded64): same idea, offsets in doubles.
struct Blob64 {
    std::vector<double> host;
    DevBuf dev;
    size_t add(const std::vector<double> &v) {
        size_t off = (host.size() + 31) & ~size_t(31);
        host.resize(off + v.size());
        for (size_t i = 0; i < v.size(); ++i) host[off + i] = v[i];
        return off;
    }
    int upload() { host.resize((host.size() + 31) & ~size_t(31)); return dev.upload(host.data(), host.size() * sizeof(double)); }
    const double *at(size_t off) const { return dev.as<double>() + off; }
};

}  // namespace dgdm
