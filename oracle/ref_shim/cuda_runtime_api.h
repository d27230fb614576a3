// Stand-in for <cuda_runtime_api.h> -- TEST INFRASTRUCTURE ONLY (see README.md): "device" memory is host memory.
#pragma once
#include <cstdlib>
#include <cstring>
typedef void *cudaStream_t;
typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2 };
inline cudaError_t cudaMalloc(void **p, size_t n) { *p = malloc(n); return *p ? 0 : 2; }
inline cudaError_t cudaMemcpy(void *d, const void *s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return 0; }
inline cudaError_t cudaMemset(void *d, int v, size_t n) { memset(d, v, n); return 0; }
inline cudaError_t cudaDeviceSynchronize() { return 0; }
