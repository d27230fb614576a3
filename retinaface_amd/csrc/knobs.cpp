// knobs.cpp -- see knobs.h.
#include "knobs.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace rf {

namespace {

#ifdef RF_PROBES
constexpr bool kProbes = true;
#else
constexpr bool kProbes = false;
#endif

constexpr int kAny = -0x7fffffff;          // `allowed` terminator
constexpr int kPresence = -0x7ffffffe;     // the knob is "set or not": any value (even empty) counts as 1

struct Spec {
    const char *name;
    bool semantic;          // exists in the product build
    int def;
    int allowed[14];        // the values the dispatch code knows, terminated by kAny; {kPresence} = presence knob; {kAny} = any integer
};

// One row per knob, in the order of enum Knob.  `allowed` is what the dispatch code in kernels.hip / engine.cpp has a case for.
const Spec kSpecs[K_COUNT] = {
    {"RF_BLEND_FP32", true, 0, {kPresence}},
    {"RF_FORCE_SCATTER", true, 0, {kPresence}},
    {"RF_SCATTER_PER_FRAME", true, 0, {kPresence}},
    {"RF_SYNC_SPLIT", true, 1, {0, 1, kAny}},
    {"RF_PREBUILD_LANES", true, 0, {0, 1, kAny}},
    {"RF_HOST_TRACE", true, 0, {kPresence}},
    {"RF_STEM2", false, 1, {0, 1, 2, 3, kAny}},
    {"RF_STEM_RAW", false, 1, {0, 1, 2, kAny}},
    {"RF_STEM2_PAD", false, 0, {0, 3, 7, kAny}},
    {"RF_STEM2_V2", false, 15, {0, 1, 2, 3, 5, 7, 15, 31, 47, 95, 127, kAny}},
    {"RF_STEM2_DC", false, 1, {0, 1, kAny}},
    {"RF_DWPWWS", false, 0, {0, 2, 3, 12, 13, kAny}},
    {"RF_DWPAD", false, 1, {0, 1, kAny}},
    {"RF_TILE_A", false, 1, {0, 1, 2, kAny}},
    {"RF_TILE_B", false, 1, {0, 1, 2, kAny}},
    {"RF_TILE_C", false, 1, {0, 1, 2, kAny}},
    {"RF_TILE_D", false, 0, {0, 1, kAny}},
    {"RF_TILE64", false, 0, {0, 1, 2, kAny}},
    {"RF_TILE128", false, -1, {-1, 0, 1, 2, 3, kAny}},
    {"RF_TILE256", false, 2, {0, 1, 2, kAny}},
    {"RF_DWPW2", false, 1, {0, 1, kAny}},
    {"RF_DWPW2_RING", false, 0, {0, 1, kAny}},
    {"RF_DWPW2_CHAIN", false, 0, {0, 1, kAny}},
    {"RF_DWPW2_HPAD", false, 1, {0, 1, kAny}},
    {"RF_DWPW2_LAY2", false, 1, {0, 1, kAny}},
    {"RF_CONV3", false, -1, {-1, 0, 1, 2, kAny}},
    {"RF_CONV3WS", false, 1, {0, 1, 22, 23, 32, 33, 122, 132, kAny}},
    {"RF_CONV3UPWS", false, 1, {0, 1, 2, 3, 12, 13, kAny}},
    {"RF_SSHTAIL", false, 1, {0, 1, 2, kAny}},
    {"RF_HEAD_START", false, 0, {0, 1, kAny}},
    {"RF_NT_COPY", false, 1, {0, 1, kAny}},
    {"RF_COPY_STREAMS", false, 2, {1, 2, kAny}},
    {"RF_CU_SPLIT", false, 0, {0, 1, kAny}},
    {"RF_WIDE_I8", false, 3, {0, 1, 2, 3, 4, 5, 6, 7, kAny}},
    {"RF_WIDE128", false, 1, {0, 1, 2, kAny}},
    {"RF_WIDE256", false, 1, {0, 1, kAny}},
};

int g_value[K_COUNT];
float g_min_rounds = 1.0f;
float g_grid_frac = 1.0f;
std::once_flag g_once;

void parse_all() {
    for (int k = 0; k < K_COUNT; k++) {
        const Spec &s = kSpecs[k];
        g_value[k] = s.def;
        const char *e = getenv(s.name);
        if (!e || s.semantic) continue;
        if (s.allowed[0] == kPresence) { g_value[k] = 1; continue; }
        char *end = nullptr;
        const long v = strtol(e, &end, 10);
        bool ok = end != e && *end == '\0';
        if (ok && s.allowed[0] != kAny) {
            ok = false;
            for (int i = 0; i < 14 && s.allowed[i] != kAny; i++) ok = ok || s.allowed[i] == (int)v;
        }
        if (!ok) {
            fprintf(stderr, "[retinaface_amd] %s=%s is not a value this knob knows: using the default %d\n", s.name, e, s.def);
            continue;
        }
        if (!s.semantic && !kProbes) {
            if ((int)v != s.def)
                fprintf(stderr, "[retinaface_amd] %s=%s ignored: probe knobs and the kernel variants they select are compiled into "
                                "libretinaface_amd_probe.so only (make probe; RETINAFACE_AMD_LIB=<that file>)\n", s.name, e);
            continue;
        }
        g_value[k] = (int)v;
    }
    if (kProbes) {
        if (const char *e = getenv("RF_PERSIST_MIN_ROUNDS")) {
            char *end = nullptr;
            const float v = strtof(e, &end);
            if (end != e && *end == '\0' && v >= 0.f && v <= 64.f) g_min_rounds = v;
            else fprintf(stderr, "[retinaface_amd] RF_PERSIST_MIN_ROUNDS=%s is not a number in [0, 64]: using the default %g\n", e, (double)g_min_rounds);
        }
        if (const char *e = getenv("RF_GRID_FRAC")) {
            char *end = nullptr;
            const float v = strtof(e, &end);
            if (end != e && *end == '\0' && v >= 0.1f && v <= 1.f) g_grid_frac = v;
            else fprintf(stderr, "[retinaface_amd] RF_GRID_FRAC=%s is not a number in [0.1, 1]: using the default %g\n", e, (double)g_grid_frac);
        }
    }
}

}  // namespace

int knob(Knob k) {
    // semantic knobs are looked up on every query (engines read them at construction / launch-build time and tests toggle them
    // between engines of one process); probe knobs are fixed for the life of the process
    if (kSpecs[k].semantic) {
        const char *e = getenv(kSpecs[k].name);
        if (!e) return kSpecs[k].def;
        if (kSpecs[k].allowed[0] == kPresence) return 1;
        return atoi(e) != 0 ? 1 : 0;
    }
    std::call_once(g_once, parse_all);
    return g_value[k];
}
float knob_persist_min_rounds() {
    std::call_once(g_once, parse_all);
    return g_min_rounds;
}
float knob_grid_frac() {
    std::call_once(g_once, parse_all);
    return g_grid_frac;
}
const char *knob_name(Knob k) { return kSpecs[k].name; }
bool probes_compiled() { return kProbes; }

}  // namespace rf
