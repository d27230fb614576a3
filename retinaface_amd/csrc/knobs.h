// knobs.h -- every RF_* environment knob of the library in ONE table (round 5; ADVICE r4: the knobs were ~30 function-local statics,
// each parsed where it was used, unknown values fell through to some variant silently).
//
// Two classes:
//   * SEMANTIC knobs change what the library does for a caller and exist in every build: RF_BLEND_FP32, RF_FORCE_SCATTER,
//     RF_PREBUILD_LANES, RF_HOST_TRACE.
//   * PROBE knobs select measured-and-rejected kernel variants / layouts for A/B measurements (DESIGN.md section 4 cites them).
//     They exist only in the probe build (`make probe` -> libretinaface_amd_probe.so, compiled with -DRF_PROBES, which also
//     contains the rejected kernels); the product library is compiled without those kernels, knob() returns the default there and
//     says so once on stderr if the variable is set to anything else.
// Every probe value is read once per process (std::call_once), checked against the values the knob knows, and an unknown value is
// reported on stderr and replaced by the default instead of selecting whatever the dispatch code falls through to.  Semantic knobs
// are looked up on every query (engines read them when they are built; tests toggle them between engines of one process).
#pragma once

namespace rf {

enum Knob {
    // ---- semantic (every build)
    K_BLEND_FP32,          // RF_BLEND_FP32: int8 aggregation convs blend in fp32 instead of packed integers (bit-identical; test knob)
    K_FORCE_SCATTER,       // RF_FORCE_SCATTER: treat every device frame as resident on another GPU (peer-copy path on a one-GPU box)
    K_SCATTER_PER_FRAME,   // RF_SCATTER_PER_FRAME: one peer copy per foreign frame (rounds 3-5) instead of one per contiguous run of frames (the split A/B of bench.py)
    K_SYNC_SPLIT,          // RF_SYNC_SPLIT=0: a synchronous host-frame call stages + uploads its frames in ONE piece (rounds 1-5) instead of pipelined pieces
    K_PREBUILD_LANES,      // RF_PREBUILD_LANES: build every lane at rf_create instead of on first use
    K_HOST_TRACE,          // RF_HOST_TRACE: per-stage host wall clock of the calls, printed when the engine is destroyed
    // ---- probe (probe build only; the product build returns the default)
    K_STEM2,               // 0 = K_a' + a separate dwpw<16,32,s2>; 1 = stem2 7x8 tiles; 2 = 7x16, 8 waves; 3 = fp16 patch
    K_STEM_RAW,            // 1 default: the int8 stem stages aligned full-width frames as raw rows by LDS-DMA (round 6); 0 = general path only; 2 = + conv0's pixel indices from a table in memory (measured: slower)
    K_STEM2_PAD,           // 0 | 3 | 7 KB of unused LDS (occupancy probe)
    K_STEM2_V2,            // bit 0 planar conv2 tile, bit 1 conv3 -> conv4 register chain, bit 2 rotated depthwise-1 map, bit 3 raw-row staging; default 15.  Measured and rejected: bit 4 conv0 pixel table from memory, bit 5 expanded conv3 fragments, bit 6 the table's LDS reads as explicit ds_read2_b32 (31, 47, 95, 127)
    K_STEM2_DC,            // 0 = stem2 tiles without DC centring (another packed image)
    K_DWPWWS,              // 0 | 2 | 3 | 12 | 13: warp-specialised / per-wave-DMA depthwise-pointwise blocks
    K_DWPAD,               // 0 = round-1 halo layout of the depthwise-pointwise blocks
    K_TILE_A, K_TILE_B, K_TILE_C, K_TILE_D,      // int8 big-map tile shapes
    K_TILE64, K_TILE128, K_TILE256,              // tile shapes of the 64 / 128 / 256-channel blocks (-1 = per-precision default)
    K_DWPW2,               // 0 = blocks 2 and 3 as two launches
    K_DWPW2_RING, K_DWPW2_CHAIN, K_DWPW2_HPAD, K_DWPW2_LAY2,
    K_CONV3,               // -1 default; 0 round-1 split without bank-row padding; 1 / 2 ALLC
    K_CONV3WS,             // 1 default (fp16: 132); 0 lock-step; 22 23 32 33 122 132
    K_CONV3UPWS,           // 1 auto; 0 lock-step; 2 3 producer wave; 12 13 per-wave DMA
    K_SSHTAIL,             // 1 default; 0 two conv3x3<16,*> launches; 2 = 3 workgroups per CU
    K_HEAD_START,          // 1: launches of <= max_batch images start their first kernel eagerly and submit the graph of the rest while it runs; 0 = one graph
    K_NT_COPY,             // 1 default: host frames are staged into pinned memory with non-temporal stores (copier.h); 0 = memcpy
    K_COPY_STREAMS,        // 1 | 2 upload streams for staged host frames
    K_CU_SPLIT,            // 0 default; 1: lane l's stream is confined to half of every XCD's CUs (hipExtStreamCreateWithCUMask), halves alternate by lane
    K_WIDE_I8,             // int8 engine on K_b(8): bit 0 = the 256-channel block, bit 1 = the plain 128-channel blocks, bit 2 = the 128-channel lateral block; default 3
    K_WIDE128,             // 1 default: the fp16 128-channel block with the fused lateral on K_b(8); 0 = none, 2 = all five 128-channel blocks
    K_WIDE256,             // 1 default: the fp16 256-channel block on the 8-wave weights-stationary kernel (round 5); 0 = K_b, matrix streamed from L2
    K_COUNT
};

int knob(Knob k);                 // validated value, read once per process
float knob_grid_frac();           // RF_GRID_FRAC (probe): persistent grids sized for this fraction of the resident workgroup slots (another lane's kernel may take the rest)
float knob_persist_min_rounds();  // RF_PERSIST_MIN_ROUNDS (probe): tiles per resident workgroup below which a persistent grid is not trimmed
const char *knob_name(Knob k);
bool probes_compiled();           // true in libretinaface_amd_probe.so

}  // namespace rf
