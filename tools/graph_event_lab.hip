// Can a kernel inside a captured hipGraph be timed with events?  hipEventRecord / hipEventRecordWithFlags(hipEventRecordExternal) during stream capture, replay, hipEventElapsedTime.
//   hipcc --offload-arch=gfx950 -O2 tools/graph_event_lab.hip -o tools/graph_event_lab && tools/graph_event_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#define P(x) do { hipError_t e_ = (x); printf("  %-70s -> %s\n", #x, hipGetErrorString(e_)); } while (0)
__global__ void spin(float* p, int n) { float v = p[threadIdx.x]; for (int i = 0; i < n; i++) v = v * 1.0001f + 0.5f; p[threadIdx.x] = v; }
int main() {
    float* d; hipMalloc(&d, 4096);
    hipStream_t s; hipStreamCreate(&s);
    hipStream_t snb; hipStreamCreateWithPriority(&snb, hipStreamNonBlocking, 0);
    hipEvent_t pe0, pe1; hipEventCreate(&pe0); hipEventCreate(&pe1);     // created before anything else happens
    for (int variant = 0; variant < 6; variant++) {
        if (variant >= 3) { s = snb; }
        printf("variant %d%s: %s\n", variant, variant >= 3 ? " (non-blocking priority stream)" : "", variant % 3 == 0 ? "hipEventRecordWithFlags(External), mode ThreadLocal" : variant % 3 == 1 ? "hipEventRecordWithFlags(External), mode Global" : "plain hipEventRecord, mode ThreadLocal");
        hipEvent_t e0, e1; if (variant == 5) { e0 = pe0; e1 = pe1; printf("  (events created at program start, recorded externally)\n"); } else { P(hipEventCreate(&e0)); P(hipEventCreate(&e1)); }
        hipGraph_t g; hipGraphExec_t ge;
        P(hipStreamBeginCapture(s, variant % 3 == 1 ? hipStreamCaptureModeGlobal : hipStreamCaptureModeThreadLocal));
        hipStreamCaptureStatus st; P(hipStreamIsCapturing(s, &st)); printf("  capture status %d\n", (int)st);
        if (variant % 3 < 2 || variant == 5) P(hipEventRecordWithFlags(e0, s, hipEventRecordExternal)); else P(hipEventRecord(e0, s));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 2000000); P(hipGetLastError());
        if (variant % 3 < 2 || variant == 5) P(hipEventRecordWithFlags(e1, s, hipEventRecordExternal)); else P(hipEventRecord(e1, s));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 10); P(hipGetLastError());
        P(hipStreamEndCapture(s, &g));
        size_t nn = 0; P(hipGraphGetNodes(g, nullptr, &nn)); printf("  graph nodes: %zu\n", nn);
        P(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; r++) { P(hipGraphLaunch(ge, s)); P(hipStreamSynchronize(s)); float ms = -1.f; P(hipEventElapsedTime(&ms, e0, e1)); printf("  replay %d: elapsed %.3f ms\n", r, ms); }
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
