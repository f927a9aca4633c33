// vnd_rccl.hpp - sharding helpers for hosts that do not go through torch.distributed: vnd_shard_range, the tap table over RCCL.
// (one translation unit: included by vnd_amd.hip after vnd_objects.hpp; everything static here is private to the library)
#pragma once

extern "C" {

// ---- sharding helpers for hosts that do not go through torch.distributed -------
vnd_status vnd_shard_range(int64_t total, int32_t world_size, int32_t rank, int64_t *first, int64_t *count)
{
    if (!first || !count) return fail(VND_ERR_INVALID, "null out pointer");
    if (total < 0 || world_size <= 0 || rank < 0 || rank >= world_size)
        return fail(VND_ERR_INVALID, "bad shard query: %lld streams, rank %d of %d", (long long)total, rank, world_size);
    const int64_t base = total / world_size, extra = total % world_size;
    *count = base + (rank < extra ? 1 : 0);
    *first = rank * base + std::min<int64_t>(rank, extra);
    return VND_OK;
}

// RCCL is loaded on first use: a host that never shards needs no librccl
namespace {
struct RcclApi {
    int (*broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*error_string)(int) = nullptr;
    bool tried = false;
};
RcclApi *rccl_api()
{
    static RcclApi api;
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (!api.tried) {
        api.tried = true;
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (h) {
            api.broadcast = (decltype(api.broadcast))dlsym(h, "ncclBroadcast");
            api.error_string = (decltype(api.error_string))dlsym(h, "ncclGetErrorString");
        }
    }
    return &api;
}
constexpr int kNcclUint8 = 1;         // rccl.h: ncclDataType_t
}  // namespace

vnd_status vnd_taps_broadcast_rccl(vnd_ctx *ctx, vnd_taps **taps, int32_t root, int32_t rank, void *rccl_comm,
                                   void *stream_)
{
    if (!ctx || !taps || !rccl_comm) return fail(VND_ERR_INVALID, "null context, table slot or communicator");
    if (rank == root && !*taps) return fail(VND_ERR_INVALID, "the root rank has no table to send");
    RcclApi &api = *rccl_api();
    if (!api.broadcast) return fail(VND_ERR_UNSUPPORTED, "librccl.so could not be loaded");
    DeviceScope on(ctx->device);
    hipStream_t stream = (hipStream_t)stream_;
    auto rccl_try = [&](int rc, const char *what) -> vnd_status {
        if (rc == 0) return VND_OK;
        return fail(VND_ERR_HIP, "%s: %s", what, api.error_string ? api.error_string(rc) : "RCCL error");
    };
    // two broadcasts: the image's length, then the image (32 B header + 8 B per tap)
    int64_t bytes = 0;
    std::vector<char> image;
    if (rank == root) {
        vnd_status st = vnd_taps_serialize(*taps, nullptr, 0, &bytes);
        if (st != VND_OK) return st;
        image.resize((size_t)bytes);
        st = vnd_taps_serialize(*taps, image.data(), bytes, &bytes);
        if (st != VND_OK) return st;
    }
    int64_t *d_len = nullptr;
    HIP_TRY(hipMalloc((void **)&d_len, sizeof(int64_t)));
    char *d_body = nullptr;
    vnd_status st = VND_OK;
    do {
        if (rank == root && hipMemcpyAsync(d_len, &bytes, sizeof bytes, hipMemcpyHostToDevice, stream) != hipSuccess) { st = fail(VND_ERR_HIP, "upload of the image length failed"); break; }
        if ((st = rccl_try(api.broadcast(d_len, d_len, sizeof(int64_t), kNcclUint8, root, rccl_comm, stream), "ncclBroadcast(length)")) != VND_OK) break;
        if (hipMemcpyAsync(&bytes, d_len, sizeof bytes, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { st = fail(VND_ERR_HIP, "download of the image length failed"); break; }
        if (bytes < 32 || bytes > ((int64_t)1 << 31)) { st = fail(VND_ERR_INVALID, "implausible tap image length %lld", (long long)bytes); break; }
        if (hipMalloc((void **)&d_body, (size_t)bytes) != hipSuccess) { st = fail(VND_ERR_NOMEM, "no device memory for the tap image"); break; }
        if (rank == root && hipMemcpyAsync(d_body, image.data(), (size_t)bytes, hipMemcpyHostToDevice, stream) != hipSuccess) { st = fail(VND_ERR_HIP, "upload of the tap image failed"); break; }
        if ((st = rccl_try(api.broadcast(d_body, d_body, (size_t)bytes, kNcclUint8, root, rccl_comm, stream), "ncclBroadcast(image)")) != VND_OK) break;
        if (rank != root) {
            image.resize((size_t)bytes);
            if (hipMemcpyAsync(image.data(), d_body, (size_t)bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) { st = fail(VND_ERR_HIP, "download of the tap image failed"); break; }
        }
        if (hipStreamSynchronize(stream) != hipSuccess) { st = fail(VND_ERR_HIP, "stream synchronisation failed"); break; }
        if (rank != root) {
            // what arrived must be a tap image of exactly the announced length before anything is built from it
            // (a communicator whose ranks disagree on the root, or a torn transfer, shows up here, loudly)
            const int32_t *hd = (const int32_t *)image.data();
            if (hd[0] != kMagic || hd[1] != VND_TAPS_IMAGE_VERSION) { st = fail(VND_ERR_INVALID, "rank %d received %lld bytes that are not a tap image (magic %08x, version %d)", rank, (long long)bytes, (unsigned)hd[0], hd[1]); break; }
            const int64_t words = 8 + ((int64_t)hd[2] + 1) + 2 * (int64_t)hd[3] + (hd[5] ? ((int64_t)hd[2] + 1) + 2 * (int64_t)hd[4] : 0) + (hd[6] ? ((int64_t)hd[2] + 3) / 4 : 0);
            if (hd[2] <= 0 || hd[3] < 0 || hd[4] < 0 || words * 4 != bytes) { st = fail(VND_ERR_INVALID, "rank %d: the tap image's header (%d channels, %d taps, %d segments) does not match its %lld bytes", rank, hd[2], hd[3], hd[4], (long long)bytes); break; }
            st = vnd_taps_deserialize(ctx, image.data(), bytes, taps);
        }
    } while (false);
    if (d_body) (void)hipFree(d_body);
    (void)hipFree(d_len);
    return st;
}

}  // extern "C"
