// RAII wrapper of the C ABI handle (include/ppcr.h) used by the C++ classes of this directory.
#pragma once
#include <stdexcept>
#include <string>

#include "ppcr.h"

namespace prob_point_cloud_registration {

class DeviceError : public std::runtime_error {
public:
    DeviceError(int code, const std::string &what) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }

private:
    int code_;
};

class DeviceContext {
public:
    explicit DeviceContext(int device_id = 0)
    {
        const int rc = ppcr_create(device_id, &ctx_);
        if (rc != PPCR_OK) throw DeviceError(rc, std::string("ppcr_create: ") + ppcr_last_error(nullptr));
    }
    ~DeviceContext() { ppcr_destroy(ctx_); }
    DeviceContext(const DeviceContext &) = delete;
    DeviceContext &operator=(const DeviceContext &) = delete;
    ppcr_ctx *get() const { return ctx_; }
    void check(int rc, const char *what) const
    {
        if (rc != PPCR_OK) throw DeviceError(rc, std::string(what) + ": " + ppcr_last_error(ctx_));
    }

private:
    ppcr_ctx *ctx_ = nullptr;
};

}  // namespace prob_point_cloud_registration
