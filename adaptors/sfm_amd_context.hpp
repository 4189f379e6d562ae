// Shared by the adaptor translation units: the device context of the calling thread.
//
// The reference calls match() and RANSAC() from its single main thread (main.cpp:94-100); the library wants
// one context per device per host thread, so each thread gets its own on first use.  Device: environment
// variable SFM_AMD_DEVICE (default 0).  The context is released when the thread ends; objects still alive on
// it keep it alive (the library counts references), so the order of destruction at exit does not matter.
#pragma once
#include <sfm_amd.h>

#include <cstdlib>
#include <new>
#include <stdexcept>
#include <string>

namespace sfm_amd_adaptor
{
    struct ContextHolder
    {
        sfm_ctx *ctx = nullptr;
        ~ContextHolder()
        {
            if (ctx) sfm_ctx_destroy(ctx);
        }
    };

    inline sfm_ctx *context()
    {
        static thread_local ContextHolder holder;
        if (!holder.ctx) {
            const char *dev = std::getenv("SFM_AMD_DEVICE");
            const int rc = sfm_ctx_create(dev ? std::atoi(dev) : 0, &holder.ctx);
            if (rc != SFM_OK) throw std::runtime_error("sfm_amd: no usable gfx950 device (sfm_ctx_create status " + std::to_string(rc) + ")");
        }
        return holder.ctx;
    }

    // Grow-only table in page-locked host memory (sfm_host_alloc): what the adaptors pack the caller's
    // vectors into, so that the upload runs at the link's rate.  One per thread and role, kept between calls.
    struct PinnedTable
    {
        void *p = nullptr;
        size_t cap = 0;
        ~PinnedTable() { sfm_host_free(p); }
        template <typename T> T *reserve(sfm_ctx *ctx, size_t count)
        {
            const size_t bytes = count * sizeof(T);
            if (bytes > cap) {
                sfm_host_free(p);
                p = nullptr;
                cap = 0;
                const size_t want = bytes + bytes / 4 + 4096;
                if (sfm_host_alloc(ctx, want, &p) != SFM_OK) throw std::bad_alloc();
                cap = want;
            }
            return static_cast<T *>(p);
        }
    };

    // Status codes back to the reference's conventions: invalid arguments are std::invalid_argument
    // (match.cpp:18, eightPoint.cpp:61,230), everything else std::runtime_error.
    [[noreturn]] inline void fail(sfm_ctx *ctx, int rc, const char *what)
    {
        std::string msg = std::string(what) + ": sfm_amd status " + std::to_string(rc);
        if (ctx && sfm_last_error(ctx)[0]) msg = sfm_last_error(ctx);
        if (rc == SFM_ERR_INVALID_ARG || rc == SFM_ERR_UNSUPPORTED) throw std::invalid_argument(msg);
        throw std::runtime_error(msg);
    }
} // namespace sfm_amd_adaptor
