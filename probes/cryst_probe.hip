// Checks the building blocks of k_cryst_fused (csrc/ltmi_cryst.hip) on the hardware: the permlane swaps
// (register index <-> lane[5:4]) and one 256-point transform against a double-precision DFT.
//   hipcc -O3 --offload-arch=gfx950 -Ilibertem_amd/csrc -Iinclude probes/cryst_probe.hip -o probes/cryst_probe
#include "../libertem_amd/csrc/ltmi_cryst.hip"
#include <vector>
#include <complex>
#include <cmath>
using namespace ltmi;
namespace ltmi { void set_error(const char *, ...) {} }   // (the library's error slot: not linked here)

__global__ void k_swap(float *io) {            // io[r][lane][2]
    const int t = threadIdx.x;
    v2f u[4];
    for (int r = 0; r < 4; ++r) u[r] = (v2f){io[(r * 64 + t) * 2], io[(r * 64 + t) * 2 + 1]};
    cf_swap_a(u);
    for (int r = 0; r < 4; ++r) { io[(r * 64 + t) * 2] = u[r].x; io[(r * 64 + t) * 2 + 1] = u[r].y; }
}

__global__ void k_fft(const float *in, float *out, int column_style) {   // in: 256 complex natural, out: Z natural
    __shared__ v2f buf[256];
    const int t = threadIdx.x;
    const int sig = cf_sigma(t);
    CfLane c;
    for (int r = 1; r < 4; ++r) {
        double s, co;
        sincospi(-2.0 * (double)((t & 15) * r) / 64.0, &s, &co);
        c.tw[0][r - 1] = (v2f){(float)co, (float)s};
        sincospi(-2.0 * (double)((t & 3) * r) / 16.0, &s, &co);
        c.tw[1][r - 1] = (v2f){(float)co, (float)s};
        sincospi(-2.0 * (double)(sig * r) / 256.0, &s, &co);
        c.tw[2][r - 1] = (v2f){(float)co, (float)s};
        for (int pi = 0; pi < 3; ++pi) c.twr[pi][r - 1] = (v2f){-c.tw[pi][r - 1].y, c.tw[pi][r - 1].x};
    }
    const int b2 = (t >> 2) & 3, b0 = t & 3;
    const int base_b = 64 * b2 + ((t & ~12) | (b2 << 2)), base_c = 64 * b0 + ((t & ~3) | b0);
    for (int r = 0; r < 4; ++r) {
        c.wB[r] = 64 * r + (t ^ (r << 2));
        c.rB[r] = base_b ^ (r << 2);
        c.wC[r] = 64 * r + (t ^ r);
        c.rC[r] = base_c ^ r;
    }
    v2f u[4];
    if (column_style) {
        for (int r = 0; r < 4; ++r) {
            const int n = 64 * r + 4 * (t & 15) + (t >> 4);
            u[r] = (v2f){in[2 * n], in[2 * n + 1]};
        }
    } else {
        for (int j = 0; j < 4; ++j) u[j] = (v2f){in[2 * (4 * t + j)], in[2 * (4 * t + j) + 1]};
        cf_swap_a(u);
    }
    cf_core(buf, c, u);
    for (int r = 0; r < 4; ++r) { out[2 * (sig + 64 * r)] = u[r].x; out[2 * (sig + 64 * r) + 1] = u[r].y; }
}

int main() {
    std::vector<float> h(512);
    for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; ++l) { h[(r * 64 + l) * 2] = 100 * r + l; h[(r * 64 + l) * 2 + 1] = -(100 * r + l); }
    float *d; hipMalloc(&d, 2048); hipMemcpy(d, h.data(), 2048, hipMemcpyHostToDevice);
    k_swap<<<1, 64>>>(d);
    std::vector<float> g(512); hipMemcpy(g.data(), d, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; ++l) {
        const int lp = (r << 4) | (l & 15), rp = l >> 4;
        if (g[(rp * 64 + lp) * 2] != h[(r * 64 + l) * 2]) ++bad;
    }
    printf("swap_a: %d of 256 elements not where the model puts them\n", bad);
    if (bad) { for (int r = 0; r < 4; ++r) { printf("reg %d:", r); for (int l = 0; l < 64; l += 4) printf(" %g", g[(r * 64 + l) * 2]); printf("\n"); } }
    std::vector<float> x(512); std::vector<std::complex<double>> X(256);
    srand(1);
    for (auto &v : x) v = (float)(rand() % 2001 - 1000);
    for (int k = 0; k < 256; ++k) { std::complex<double> s = 0; for (int n = 0; n < 256; ++n) s += std::complex<double>(x[2 * n], x[2 * n + 1]) * std::polar(1.0, -2 * M_PI * k * n / 256.0); X[k] = s; }
    float *di, *dout; hipMalloc(&di, 2048); hipMalloc(&dout, 2048); hipMemcpy(di, x.data(), 2048, hipMemcpyHostToDevice);
    for (int style = 0; style < 2; ++style) {
        k_fft<<<1, 64>>>(di, dout, style);
        std::vector<float> z(512); hipMemcpy(z.data(), dout, 2048, hipMemcpyDeviceToHost);
        double err = 0, nrm = 0;
        for (int k = 0; k < 256; ++k) { err = std::max(err, std::abs(std::complex<double>(z[2 * k], z[2 * k + 1]) - X[k])); nrm = std::max(nrm, std::abs(X[k])); }
        printf("%s transform: max error %.3g of max |Z| %.3g\n", style ? "column-style" : "row-style", err, nrm);
    }
    return 0;
}
