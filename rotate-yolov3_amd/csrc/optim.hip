// rotate-yolov3_amd/csrc/optim.hip -- one launch for the whole SGD step (train.py:70-83 builds torch.optim.SGD with
// momentum + nesterov, weight decay on the conv weights only; ATen runs it as ~30 foreach kernels over 222 tensors).
// Same fp32 arithmetic, same order (built without fp contraction):
//   d = g (+ wd * p);  buf = first ? d : momentum * buf + d;  d = nesterov ? d + momentum * buf : buf;  p -= lr * d
// Optional per-group gradient scale (hparam slot 3; 0 = none): data-parallel training passes 1/world here instead of dividing
// the 250 MB all-reduced gradient in a separate pass (dist.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ryolo.h"

namespace {

__global__ void __launch_bounds__(256) sgd_batch_kernel(const ryolo_sgd_job *__restrict__ jobs, int njobs,
                                                        const float *__restrict__ hp /* [groups][4] lr, momentum, wd, grad scale */,
                                                        int nesterov) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const ryolo_sgd_job j = jobs[lo];
    const float lr = hp[j.group * 4 + 0], mom = hp[j.group * 4 + 1], wd = hp[j.group * 4 + 2], gs = hp[j.group * 4 + 3];
    float *__restrict__ p = (float *)j.p;
    const float *__restrict__ g = (const float *)j.g;
    float *__restrict__ buf = (float *)j.buf;
    const long long nblk = j.block_end - j.block_begin;
    for (long long i = (long long)((int)blockIdx.x - j.block_begin) * 256 + threadIdx.x; i < j.n; i += nblk * 256) {
        const float pv = p[i];
        float d = g[i];
        if (gs != 0.f) d = d * gs;
        if (wd != 0.f) d = d + wd * pv;
        if (mom != 0.f) {
            const float b = j.first ? d : mom * buf[i] + d;
            buf[i] = b;
            d = nesterov ? d + mom * b : b;
        }
        p[i] = pv - lr * d;
    }
}

}  // namespace

extern "C" {

int ryolo_sgd_step(const ryolo_sgd_job *device_jobs, int njobs, int total_blocks, const float *group_hparams, int nesterov,
                   void *stream) {
    if (!device_jobs || njobs <= 0 || total_blocks <= 0 || !group_hparams) return RYOLO_EINVAL;
    hipLaunchKernelGGL(sgd_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, device_jobs, njobs,
                       group_hparams, nesterov);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

}  // extern "C"
