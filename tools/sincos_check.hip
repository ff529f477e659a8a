// GPU check of the fp64 sincos used for body angles (mgx_sim.h r_sincos<double>) against long double libm: build with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/sincos_check tools/sincos_check.hip && tools/sincos_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../magical_amd/csrc/mgx_sim.h"
__global__ void k(const double *a, double *s, double *c, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) mgx::r_sincos<double>(a[i], s[i], c[i]); }
int main() {
    const int n = 1 << 20; double *a, *s, *c;
    hipMallocManaged(&a, n * 8); hipMallocManaged(&s, n * 8); hipMallocManaged(&c, n * 8);
    srand(1);
    for (int i = 0; i < n; i++) { double u = rand() / (double)RAND_MAX; a[i] = (i % 3 == 0 ? 6.3 : (i % 3 == 1 ? 200.0 : 5000.0)) * (2 * u - 1); if (i < 64) a[i] = (i - 32) * M_PI / 4; }
    k<<<n / 256, 256>>>(a, s, c, n); hipDeviceSynchronize();
    double ms = 0, mc = 0; int bad = 0;
    for (int i = 0; i < n; i++) {
        long double rs = sinl((long double)a[i]), rc = cosl((long double)a[i]);
        double us = std::fabs((double)((s[i] - rs) / (long double)std::ldexp(1.0, std::ilogb((double)rs) - 52)));
        double uc = std::fabs((double)((c[i] - rc) / (long double)std::ldexp(1.0, std::ilogb((double)rc) - 52)));
        if (std::fabs((double)rs) > 1e-8 && us > ms) ms = us; if (std::fabs((double)rc) > 1e-8 && uc > mc) mc = uc;
        if (std::fabs(s[i] - (double)rs) > 1e-15 || std::fabs(c[i] - (double)rc) > 1e-15) bad++;
    }
    printf("max ulp error sin %.3f cos %.3f; abs > 1e-15: %d\n", ms, mc, bad);
}
