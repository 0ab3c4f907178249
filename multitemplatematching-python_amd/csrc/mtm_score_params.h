// Host-visible parameters of the score-map kernels in mtm_k_score.hip.h (pack geometry of the dot4 kernel, tile sizes of
// the float64 kernel, the slab combination's parameter block): the placement code packs to them, the launcher launches.
#pragma once
#include <cstdint>

#include "mtm_kernels.h"
#include "../../include/mtm_hip.h"

namespace mtm {

constexpr int kDotChunk = 64;
constexpr int kDotPadRows = 3;                                   // supports PY <= 4
constexpr int kDotPackRows = kDotChunk + 2 * kDotPadRows;
constexpr int kDotChunkBytes = kDotPackRows * kDotChunk;

struct DotParams {
    const uint8_t* img;     // planar padded u8
    int pitch;              // bytes
    long long plane;
    int chans;
    int h, w;               // template size of this class
    int oh, ow;
    int ncy, ncx;           // chunk grid of the packed templates
    int n_list;             // templates in this launch
    int ntx, nty;           // output tile grid
    int nchunks;            // ceil(n_list / NT)
    int n_work;             // ntx * nty * nchunks
    int method;
    double* sumsq_out;      // MASKSQ: destination plane (pitch = st.pitch) of sum I^2 * M
};


constexpr int kF64ChunkH = 16, kF64ChunkW = 32;
constexpr int kF64BX = 128, kF64BY = 8;
constexpr int kF64LdsPitch = kF64BX + kF64ChunkW + 4;   // floats, multiple of 4


struct SlabParams {
    mtm_hit* cand_hits;
    unsigned long long* cand_counter;
    unsigned long long cand_cap;
    float cand_thr;
    int cand_min, cand_on, hits_only;
    int w, h, chans;
    const int* raw;
    long long raw_slab;        // ints per slab block: n_list * raw_map
    long long raw_map;         // ints per template map: oh * pitch
    int n_slabs;
    int oh, ow, pitch;
    int n_list;
    int method;
    int ext_on, ext_pad_;            // fused global extremum (N_object == 1): keys to ext_best, no maps, no candidates
    unsigned long long* ext_best;    // [2 * template + cand_min], as the score kernels' EXT epilogues
};

}  // namespace mtm
