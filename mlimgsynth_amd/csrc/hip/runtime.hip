// Device runtime plumbing of the kernel shim: memory, streams, events, graphs.
// Replaces the ggml-backend data-movement call sites of the reference
// (ggml_backend_tensor_set/get: src/localtensor.h:96-106, src/mlblock.c:257,
//  src/unet.c:375-384) and ggml_backend_init_* (src/mlimgsynth.c:1131-1161).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "common.hpp"

static thread_local char g_err[512] = "";
// "dry" mode: device memory calls are served from host memory so that the host-side plan builder
// (shapes, parameter names, FLOP accounting) can be exercised on a machine without a GPU.  No kernel
// can be launched in this mode: every launcher fails loudly.
static int g_dry = 0;

extern "C" {

int mlsd_set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int mlsd_check_launch(const char* what)
{
    if (g_dry) return mlsd_set_error(-2, "%s: no GPU (dry mode builds plans only)", what);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return mlsd_set_error(-(int)e - 1000, "launch %s failed: %s", what, hipGetErrorString(e));
    return 0;
}

MLSD_API const char* mlsd_last_error(void) { return g_err; }
/* diagnostics: the runtime's pending (sticky) error without clearing it, "" if none */
MLSD_API const char* mlsd_peek_runtime_error(void)
{
    if (g_dry) return "";
    hipError_t e = hipPeekAtLastError();
    return e == hipSuccess ? "" : hipGetErrorString(e);
}
MLSD_API void mlsd_runtime_dry(int on) { g_dry = on; }
MLSD_API int mlsd_runtime_is_dry(void) { return g_dry; }

MLSD_API int mlsd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

MLSD_API int mlsd_device_set(int dev)
{
    MLSD_HIP_TRY(hipSetDevice(dev));
    return 0;
}

// name: buffer of >=256 bytes. Returns 0 and fills fields.
MLSD_API int mlsd_device_info(int dev, char* name, int name_len, char* arch, int arch_len, int* n_cu,
                              size_t* mem_total, size_t* mem_free)
{
    hipDeviceProp_t p;
    MLSD_HIP_TRY(hipGetDeviceProperties(&p, dev));
    if (name) snprintf(name, name_len, "%s", p.name);
    if (arch) snprintf(arch, arch_len, "%s", p.gcnArchName);
    if (n_cu) *n_cu = p.multiProcessorCount;
    size_t f = 0, t = 0;
    int cur = 0;
    MLSD_HIP_TRY(hipGetDevice(&cur));
    if (cur != dev) MLSD_HIP_TRY(hipSetDevice(dev));
    MLSD_HIP_TRY(hipMemGetInfo(&f, &t));
    if (cur != dev) MLSD_HIP_TRY(hipSetDevice(cur));
    if (mem_total) *mem_total = t;
    if (mem_free) *mem_free = f;
    return 0;
}

MLSD_API int mlsd_malloc(void** out, size_t nbytes)
{
    *out = NULL;
    if (g_dry) { *out = calloc(1, nbytes ? nbytes : 16); return *out ? 0 : mlsd_set_error(-1, "out of host memory"); }
    MLSD_HIP_TRY(hipMalloc(out, nbytes ? nbytes : 16));
    return 0;
}

MLSD_API int mlsd_free(void* p)
{
    if (g_dry) { free(p); return 0; }
    if (p) MLSD_HIP_TRY(hipFree(p));
    return 0;
}

MLSD_API int mlsd_host_alloc(void** out, size_t nbytes)
{
    *out = NULL;
    if (g_dry) { *out = calloc(1, nbytes ? nbytes : 16); return *out ? 0 : mlsd_set_error(-1, "out of host memory"); }
    MLSD_HIP_TRY(hipHostMalloc(out, nbytes ? nbytes : 16, hipHostMallocDefault));
    return 0;
}

MLSD_API int mlsd_host_free(void* p)
{
    if (g_dry) { free(p); return 0; }
    if (p) MLSD_HIP_TRY(hipHostFree(p));
    return 0;
}

MLSD_API int mlsd_memset(void* dst, int value, size_t nbytes, void* stream)
{
    if (g_dry) { memset(dst, value, nbytes); return 0; }
    MLSD_HIP_TRY(hipMemsetAsync(dst, value, nbytes, (hipStream_t)stream));
    return 0;
}

// kind: 0 = host->device, 1 = device->host, 2 = device->device
// device -> device copies run as a kernel: hipMemcpyAsync may hand them to the copy engine, where they queue behind its current job (with streamed weights a 500 MB
// upload: 10 ms gaps between evaluations, MLSD_ENGINE_TRACE); latents and states are a few MB
__global__ void copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

MLSD_API int mlsd_memcpy(void* dst, const void* src, size_t nbytes, int kind, void* stream)
{
    if (g_dry) { memcpy(dst, src, nbytes); return 0; }
    if (kind == 2 && nbytes && !(nbytes & 15) && !(((uintptr_t)dst | (uintptr_t)src) & 15)) {
        const size_t n16 = nbytes >> 4;
        const unsigned blocks = (unsigned)(n16 >= (size_t)2048 * 256 ? 2048 : (n16 + 255) / 256);
        hipLaunchKernelGGL(copy16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, n16);
        return mlsd_check_launch("copy16_kernel");
    }
    hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    MLSD_HIP_TRY(hipMemcpyAsync(dst, src, nbytes, k, (hipStream_t)stream));
    return 0;
}

MLSD_API int mlsd_stream_create(void** out)
{
    hipStream_t s;
    if (g_dry) { *out = (void*)(uintptr_t)0xD0;  return 0; }      /* dry runtime: a token, never handed to HIP */
    MLSD_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *out = (void*)s;
    return 0;
}

/* A stream whose kernels may only run on a subset of the CUs (hipExtStreamCreateWithCUMask): `mask` holds n_words x 32 bits.
 * Used to run two independent sub-batches side by side on two halves of the chip (DESIGN.md "two partitions"). */
MLSD_API int mlsd_stream_create_masked(void** out, const uint32_t* mask, int n_words)
{
    hipStream_t s;
    MLSD_HIP_TRY(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask));
    *out = (void*)s;
    return 0;
}

/* census: which XCD (HW_REG_XCC_ID) and CU the blocks of a launch on `stream` land on; out[block] = (xcc << 16) | (se << 8) | cu */
__global__ void census_kernel(unsigned* out, int spin)
{
    if (threadIdx.x == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));     /* HW_REG_XCC_ID bits [3:0] */
        unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    /* HW_REG_HW_ID */
        out[blockIdx.x] = (xcc << 16) | (hwid & 0xffff);
        const long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < spin) { }
    }
}
MLSD_API int mlsd_cu_census(unsigned* out_dev, int n_blocks, int spin_cycles, void* stream)
{
    hipLaunchKernelGGL(census_kernel, dim3(n_blocks), dim3(64), 65536, (hipStream_t)stream, out_dev, spin_cycles);
    return mlsd_check_launch("census_kernel");
}

MLSD_API int mlsd_stream_destroy(void* s)
{
    if (g_dry || s == (void*)(uintptr_t)0xD0) return 0;
    if (s) MLSD_HIP_TRY(hipStreamDestroy((hipStream_t)s));
    return 0;
}

MLSD_API int mlsd_stream_sync(void* s)
{
    if (g_dry) return 0;
    MLSD_HIP_TRY(hipStreamSynchronize((hipStream_t)s));
    return 0;
}

MLSD_API int mlsd_device_sync(void)
{
    MLSD_HIP_TRY(hipDeviceSynchronize());
    return 0;
}

MLSD_API int mlsd_event_create(void** out)
{
    hipEvent_t e;
    if (g_dry) { *out = (void*)(uintptr_t)0xE0; return 0; }
    MLSD_HIP_TRY(hipEventCreate(&e));
    *out = (void*)e;
    return 0;
}

MLSD_API int mlsd_event_destroy(void* e)
{
    if (g_dry || e == (void*)(uintptr_t)0xE0) return 0;
    if (e) MLSD_HIP_TRY(hipEventDestroy((hipEvent_t)e));
    return 0;
}

MLSD_API int mlsd_event_record(void* e, void* stream)
{
    MLSD_HIP_TRY(hipEventRecord((hipEvent_t)e, (hipStream_t)stream));
    return 0;
}

MLSD_API int mlsd_event_sync(void* e)
{
    MLSD_HIP_TRY(hipEventSynchronize((hipEvent_t)e));
    return 0;
}

MLSD_API int mlsd_event_elapsed_ms(void* e0, void* e1, float* ms)
{
    MLSD_HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)e0, (hipEvent_t)e1));
    return 0;
}

MLSD_API int mlsd_stream_wait_event(void* stream, void* e)
{
    MLSD_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)e, 0));
    return 0;
}

// ---- hipGraph capture of a launch sequence (one UNet evaluation) ----------
MLSD_API int mlsd_capture_begin(void* stream)
{
    MLSD_HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return 0;
}

MLSD_API int mlsd_capture_end(void* stream, void** graph_exec)
{
    hipGraph_t g = NULL;
    *graph_exec = NULL;
    MLSD_HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &g));
    hipGraphExec_t ge = NULL;
    hipError_t e = hipGraphInstantiate(&ge, g, NULL, NULL, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return mlsd_set_error(-(int)e - 1000, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    *graph_exec = (void*)ge;
    return 0;
}

MLSD_API int mlsd_graph_launch(void* graph_exec, void* stream)
{
    MLSD_HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return 0;
}

MLSD_API int mlsd_graph_destroy(void* graph_exec)
{
    if (graph_exec) MLSD_HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return 0;
}

}  // extern "C"
