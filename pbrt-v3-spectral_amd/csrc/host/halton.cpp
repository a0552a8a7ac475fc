// halton.cpp -- host-side tables for the Halton sampler: the first n primes, their
// prefix sums, and the random digit permutations. The permutations are a pure
// function of the default-seeded PCG32 stream, generated base by base in prime
// order (ComputeRadicalInversePermutations, src/core/lowdiscrepancy.cpp:2490-2504;
// Shuffle, src/core/sampling.h:150-157; RNG, src/core/rng.h:61-144), so a prefix
// of the reference's 3 682 913-entry table is reproduced without storing it.
#include "scene.h"

namespace mipt {
namespace {
#include "sobol_tables.inc"

struct PCG32 {  // rng.h:61-144
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    uint32_t UniformUInt32() {
        uint64_t oldstate = state;
        state = oldstate * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
        uint32_t rot = (uint32_t)(oldstate >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    uint32_t UniformUInt32(uint32_t b) {
        uint32_t threshold = (~b + 1u) % b;
        while (true) {
            uint32_t r = UniformUInt32();
            if (r >= threshold) return r % b;
        }
    }
};

}  // namespace

void ComputeHaltonTables(int nDims, std::vector<int32_t> *primes, std::vector<int32_t> *primeSums,
                         std::vector<uint16_t> *perms) {
    primes->clear();
    primeSums->clear();
    perms->clear();
    // first nDims primes by trial division (the reference tabulates 1000, lowdiscrepancy.cpp:40-123)
    for (int c = 2; (int)primes->size() < nDims; ++c) {
        bool isPrime = true;
        for (int p : *primes) {
            if (p * p > c) break;
            if (c % p == 0) { isPrime = false; break; }
        }
        if (isPrime) primes->push_back(c);
    }
    int sum = 0;
    for (int i = 0; i < nDims; ++i) { primeSums->push_back(sum); sum += (*primes)[i]; }
    perms->resize(sum);
    PCG32 rng;
    uint16_t *p = perms->data();
    for (int i = 0; i < nDims; ++i) {
        int count = (*primes)[i];
        for (int j = 0; j < count; ++j) p[j] = (uint16_t)j;
        for (int j = 0; j < count; ++j) {  // Shuffle(p, count, 1, rng)
            int other = j + (int)rng.UniformUInt32((uint32_t)(count - j));
            std::swap(p[j], p[other]);
        }
        p += count;
    }
}

// Sampler "sobol" (sobol.h:51-62): resolution = RoundUpPow2(max extent of the sample bounds); the generator matrices of the
// first nDims dimensions and the two pixel-index matrices of that resolution (SobolIntervalToIndex, lowdiscrepancy.h:229-249).
bool ComputeSobolTables(int extent, int nDims, HostScene *scene, int *resolution, int *log2Resolution) {
    int res = 1, lg = 0;
    while (res < extent) { res *= 2; ++lg; }
    *resolution = res;
    *log2Resolution = lg;
    if (lg > kSobolResolutions || nDims > kSobolDims) return false;
    scene->sobolMatrices.assign(kSobolMatrices32, kSobolMatrices32 + (size_t)nDims * kSobolMatrixSize);
    scene->sobolVdc.assign(kSobolMatrixSize, 0);
    scene->sobolVdcInv.assign(kSobolMatrixSize, 0);
    if (lg > 0) {
        scene->sobolVdc.assign(kVdCSobolMatrices[lg - 1], kVdCSobolMatrices[lg - 1] + kSobolMatrixSize);
        scene->sobolVdcInv.assign(kVdCSobolMatricesInv[lg - 1], kVdCSobolMatricesInv[lg - 1] + kSobolMatrixSize);
    }
    return true;
}

}  // namespace mipt
