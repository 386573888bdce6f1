/*
 * rng_thrust_harness.cpp -- TEST INFRASTRUCTURE (golden-vector generator, authoring container only).
 *
 * The reference seeds `thrust::default_random_engine` and draws through
 * `thrust::uniform_real_distribution<float>(0,1)` (src/pathtrace.cu:41-45,105-106,
 * src/interactions.h:12-17).  thrust is a third-party dependency that is NOT vendored in the
 * reference (it came with the CUDA toolkit of the day, unpinned).  This image ships rocThrust
 * (ROCm 7.2.0, /opt/rocm/include/thrust), whose random/ subtree is the unchanged thrust 1.x code.
 * This program runs THAT library (host side) to emit golden streams:
 *     stdin : "<n_seeds> <n_draws>" then n_seeds unsigned seeds
 *     stdout: per seed one line: seed, then n_draws u01 values as hex bit patterns
 * Build: hipcc --offload-host-only -O2 -ffp-contract=off rng_thrust_harness.cpp -o _ref/rng_thrust
 */
#include <thrust/random.h>
#include <cstdio>
#include <cstring>

int main() {
    int ns = 0, nd = 0;
    if (scanf("%d %d", &ns, &nd) != 2) return 1;
    for (int i = 0; i < ns; ++i) {
        unsigned seed = 0;
        if (scanf("%u", &seed) != 1) return 1;
        int h = (int)seed; /* pathtrace.cu:43 stores the hash in an int */
        thrust::default_random_engine rng(h);
        thrust::uniform_real_distribution<float> u01(0, 1);
        printf("%u", seed);
        for (int k = 0; k < nd; ++k) {
            float u = u01(rng);
            unsigned bits;
            memcpy(&bits, &u, 4);
            printf(" %08x", bits);
        }
        printf("\n");
    }
    return 0;
}
