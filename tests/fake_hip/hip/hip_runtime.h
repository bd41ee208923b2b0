// tests/fake_hip/hip/hip_runtime.h — a FAKE HIP runtime header for the CPU sanitizer job (tests/test_fake_hip_cpu.py).
// TEST INFRASTRUCTURE ONLY: dsdtm_amd/csrc/api.cpp (the host side of the C ABI: stream rings, pair counters, recover slots,
// the sharded / streamed entries, frames) is compiled against THIS header with g++ -fsanitize=address,undefined (or thread)
// and linked with fake_hip.cpp, which keeps "device" memory in the host heap, queues every asynchronous operation per
// stream and executes it only when something waits for it — so a copy whose source or destination was freed too early, a
// record that points into a regrown staging buffer, or a launch accounted to the wrong stream shows up on the CPU.
// Nothing here is used by the product; the names and signatures are the subset of HIP that api.cpp and kernels.h use.
#pragma once
#include <stddef.h>
#include <stdint.h>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
typedef struct fake_stream* hipStream_t;
typedef struct fake_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocMapped = 2 };
enum hipDeviceAttribute_t { hipDeviceAttributeNumberOfXccs = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char gcnArchName[256]; int multiProcessorCount; };
struct hipPointerAttribute_t { hipMemoryType type; int device; };

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int d);
hipError_t hipDeviceSynchronize();
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned flags);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p);
hipError_t hipMemset(void* p, int v, size_t bytes);
hipError_t hipMemsetAsync(void* p, int v, size_t bytes, hipStream_t s);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind k, hipStream_t s);
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind k, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* st);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
