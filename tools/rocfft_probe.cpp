// probe: rocFFT plan-creation and execution cost for awkward even real lengths (double precision)
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    rocfft_setup();
    std::vector<size_t> lens = {4096, 5296, 5298, 5300, 5302, 6002, 6874, 7918, 8190, 8194, 9998, 5296};
    size_t batch = 64;
    double *in; double2 *out;
    hipMalloc(&in, sizeof(double) * 16384 * batch);
    hipMalloc(&out, sizeof(double2) * 8200 * batch);
    hipMemset(in, 0, sizeof(double) * 16384 * batch);
    for (size_t L : lens) {
        double t0 = now();
        rocfft_plan plan = nullptr;
        rocfft_plan_description desc = nullptr;
        rocfft_plan_description_create(&desc);
        size_t lengths[1] = {L};
        rocfft_status s = rocfft_plan_create(&plan, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                                             rocfft_precision_double, 1, lengths, batch, nullptr);
        size_t wsz = 0;
        rocfft_plan_get_work_buffer_size(plan, &wsz);
        void* wbuf = nullptr;
        rocfft_execution_info info = nullptr;
        rocfft_execution_info_create(&info);
        if (wsz) { hipMalloc(&wbuf, wsz); rocfft_execution_info_set_work_buffer(info, wbuf, wsz); }
        double t1 = now();
        void* ib[1] = {in}; void* ob[1] = {out};
        rocfft_execute(plan, ib, ob, info);
        hipDeviceSynchronize();
        double t2 = now();
        for (int i = 0; i < 20; i++) rocfft_execute(plan, ib, ob, info);
        hipDeviceSynchronize();
        double t3 = now();
        printf("L=%zu status=%d plan %.1f ms first-exec %.1f ms exec %.1f us/batch(64) work %zu B\n", L, (int)s,
               1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e6 * (t3 - t2) / 20, wsz);
        rocfft_plan_destroy(plan);
        rocfft_execution_info_destroy(info);
        if (wbuf) hipFree(wbuf);
    }
    rocfft_cleanup();
    return 0;
}
