/* TEST INFRASTRUCTURE: the handful of HIP declarations csrc/defer.h / defer.hip use, so that the slab-thread layer can
 * be built with g++ -fsanitize=thread and driven on the CPU (tests/san/defer_tsan.cpp implements them as loggers). */
#pragma once
#include <cstddef>
typedef struct stub_stream *hipStream_t;
typedef struct stub_event *hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault } hipMemcpyKind;
hipError_t hipSetDevice(int);
hipError_t hipGetLastError(void);
hipError_t hipMemcpyAsync(void *, const void *, size_t, hipMemcpyKind, hipStream_t);
hipError_t hipMemcpyPeerAsync(void *, int, const void *, int, size_t, hipStream_t);
hipError_t hipMemcpy2DAsync(void *, size_t, const void *, size_t, size_t, size_t, hipMemcpyKind, hipStream_t);
hipError_t hipMemsetAsync(void *, int, size_t, hipStream_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned);
hipError_t hipStreamSynchronize(hipStream_t);
#define hipLaunchKernelGGL(...) ((void)0)
