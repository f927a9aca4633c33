// vnd_spec.hpp - per-table specialisation of the fast-mode kernel with hipRTC.
//
// A tap table is tiny, immutable and reused across whole signals and batches, so the throughput mode
// compiles a kernel FOR it (vnd_spec_kernel.inc): tap offsets become ds_read immediates and weights
// literals, the workgroups are persistent over spans of tiles and keep the window in an LDS ring.
// Everything here is host code: the prologue generator (pure, testable without a device), the
// geometry choice, and the hipRTC / module plumbing.  The generic conv_fast_kernel remains the
// in-product fallback whenever the specialised kernel does not apply or cannot be built (no
// hipRTC, compile error, shapes outside its scope) - never a CPU path.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <stdint.h>

#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

namespace vnd {

static const char kSpecKernelSource[] =
#include "vnd_spec_kernel.inc"
    ;

struct SpecConfig {
    int nt = 256;      // threads per workgroup
    int rr = 2;        // frame pairs per lane and tile
    int pp = 4;        // ring slots
    int dd = 1;        // tiles prefetched ahead
    int la = 8;        // LDS reads in flight ahead of the FMAs
    int nt_stores = 0; // non-temporal output stores
    int exact = 0;     // VND_MODE_EXACT arithmetic: table order, separately rounded products and sums
    int epi = 0;       // exact mode, stereo: VelvetNoise.decorrelate's pointwise steps in the store phase
    int bc = 0;        // fan-out of a mono input through a stereo table: one LDS plane
    int shift = 0;     // exact mode: a second copy of every plane, one frame ahead, so that odd offsets are aligned pairs
    // WINDOW form (vnd_win.hpp): win = consecutive output frames per lane (0: the pair-read kernel above),
    // win_g = entries one base register reaches, win_lds = its LDS footprint (the halo is the table's)
    int win = 0, win_g = 0, win_lds = 0, win_per_cu = 0;    // (win_per_cu: workgroups a CU holds - LDS and registers)
    int win_xpose = 0; // the store phase transposes through LDS as interleaved frame pairs (1) or as planar chunks (0: fewer registers)
    int win_q = 0;     // window form on channel QUADS (1: signals of 4k channels, a quarter of the workgroup's lanes per channel) or OCTETS (2: 8k channels, an eighth)
    int win_s = 0;     // window form with the waves SPLIT over the two channels of a stereo signal
    int adds = 0;      // window form, fast mode: adds inside a run of equal |w|, the gain ratio once where |w| changes (vnd_win.hpp: win_adds_ok)
    int tile() const { return win ? (win_s ? nt / 2 : (win_q ? nt / (4 * win_q) : nt)) * win : 2 * nt * rr; }
    size_t lds_bytes() const
    {
        if (win) return (size_t)win_lds;
        const size_t pl = (size_t)pp * tile() + 2 * nt + (shift ? 2 : 0);
        return (bc ? 1 : 2) * (shift ? 2 : 1) * pl * 4 + (size_t)2 * (nt / 64) * rr * 2 * 4;
    }
    bool operator<(const SpecConfig &o) const
    {
        return std::tie(nt, rr, pp, dd, la, nt_stores, exact, epi, bc, shift, win, win_g, win_xpose, win_q, win_s, adds) <
               std::tie(o.nt, o.rr, o.pp, o.dd, o.la, o.nt_stores, o.exact, o.epi, o.bc, o.shift, o.win, o.win_g, o.win_xpose, o.win_q, o.win_s, o.adds);
    }
};

// the table as the generator needs it: effective weights (segment gain folded in), any order
struct SpecTable {
    int C = 0;
    std::vector<int32_t> tap_off;   // [C + 1]
    std::vector<int32_t> idx;
    std::vector<float> w;           // fast mode: weight * segment gain
    int max_index = 0;
    // exact mode: the table as the reference consumes it (table order, raw weights, segments)
    std::vector<float> w_raw;
    bool has_seg = false, apply_gain = false;
    std::vector<int32_t> seg_off, seg_end;      // per channel CSR of segments; exclusive tap ends
    std::vector<float> seg_gain;
};

// ---- run-time switches ---------------------------------------------------------------------------------------
// Two kinds.  HOST switches (VND_SPEC, VND_SPEC_EXACT, VND_SPEC_CACHE_DIR, VND_SPEC_DUMP, VND_SPEC_VERBOSE,
// VND_HOST_DIRECT, VND_HOST_TIME_PIECES / _CHUNKS) are read where they act, always.  TUNING variables - geometry overrides, A/B
// switches of the sweep tools and the tests, diagnosis builds - exist only in a TUNING SESSION: a process started with VND_TUNING=1
// reads them live at every launch plan; any other process never looks at them (spec_env returns the default).  The one list of them:
static const char *const kTuningNames[] = {
    // geometry of the per-table kernels
    "VND_SPEC_NT", "VND_SPEC_RR", "VND_SPEC_DD", "VND_SPEC_LA", "VND_SPEC_SHIFT", "VND_SPEC_QUAD_STORES", "VND_WIN_M", "VND_WIN_G", "VND_WIN_QUAD_M",
    "VND_WIN_SPLIT_LATE", "VND_WIN_SPLIT_SMALL_NT", "VND_WIN_TAIL",
    // which form runs
    "VND_WIN_EXACT", "VND_WIN_SPLIT_CLASS", "VND_WIN_QUAD", "VND_WIN_OCTET", "VND_WIN_WIDE", "VND_WIN_SPLIT", "VND_WIN_SPLIT_FANOUT", "VND_WIN_FANOUT_EPI", "VND_WIN_XPOSE_PAIRS", "VND_WIN_FAR_FIRST", "VND_WIN_ADDS", "VND_WIN_EXACT_MERGED",
    "VND_WIN_SOURCE_FANOUT", "VND_WIN_SOURCE_EPI", "VND_EPI_BLOCK_SUMS", "VND_EPI_SUMS_ONLY", "VND_EPI_WIDE", "VND_EPI_SEQ_SPLIT", "VND_EPI_PAR_COALESCED",
    // one-round launches, pacing, priorities, cache policies
    "VND_WIN_CHUNKS", "VND_WIN_BALANCE", "VND_WIN_CHUNK_LEN0", "VND_WIN_STAGGER_TICKS", "VND_WIN_PACE", "VND_WIN_PACE_MIN_TILES", "VND_WIN_PRIO", "VND_SPEC_LOAD_AUX",
    "VND_SPEC_STORE_AUX", "VND_NT_MIN_MB", "VND_NO_NT", "VND_FORCE_NT",
    // diagnosis builds
    "VND_WIN_STAMPS", "VND_WIN_STAMP_PHASES", "VND_WIN_STAMP_WAVE",
};

inline bool spec_tuning()
{
    static const bool on = [] { const char *e = getenv("VND_TUNING"); return e && *e && *e != '0'; }();
    return on;
}

// a name the library asked for that is not in the list: a library bug, reported - never by ending the host process - as
// VND_ERR_INVALID by the entry point that planned the launch (tuning_status() in vnd_plan.hpp), with the name in vnd_last_error
inline std::atomic<const char *> &spec_unregistered_name()
{
    static std::atomic<const char *> name{nullptr};
    return name;
}

inline int spec_env(const char *name, int fallback)
{
    if (!spec_tuning()) return fallback;
    bool known = false;
    for (const char *k : kTuningNames) known |= strcmp(k, name) == 0;
    if (!known) { spec_unregistered_name().store(name); return fallback; }
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : fallback;
}

// a HOST switch: read once per process (getenv is not safe against a concurrent setenv, and a host sets these before it starts)
inline int host_env_once(const char *name, int fallback, int *slot)
{
    if (*slot == INT32_MIN) { const char *e = getenv(name); *slot = (e && *e) ? atoi(e) : fallback; }
    return *slot;
}

// Smallest ring that holds one tile's window (tile + halo) plus the slot being refilled.
// Returns false when no supported geometry fits (the caller then uses the generic kernel).
inline bool spec_pick_config(const SpecTable &t, size_t lds_limit, int rr_hint, int dd_hint, SpecConfig *out,
                             bool small_tiles = false, bool bc = false, bool shift_wanted = false)
{
    const int reach = (t.max_index | 1) + 1;          // frames past a pair's first frame that an (aligned) read touches
    // (threads, pairs per lane), best first.  Measured on cfg2 and cfg3 (tools/closed/spec_try.py, several boxes):
    // the geometries land within 7 % of each other - the kernel runs on the board's power cap - with
    // 3-wave workgroups of 1536-frame tiles ahead (3 ring slots, 4 workgroups per CU).
    // Short spans (a ring is filled once per span, three tiles of loads before the first output) do
    // better with 1024-frame tiles: small_tiles starts the list there.
    static const int kLong[][2] = {{192, 4}, {256, 4}, {128, 4}, {256, 2}, {128, 2}, {256, 1}};
    static const int kShort[][2] = {{128, 4}, {256, 2}, {128, 2}, {256, 1}, {192, 4}, {256, 4}};
    // exact mode with the shifted plane copies (tools/closed/exact_geometry_try.py, cfg2): 256 threads x 2 pairs (1024-frame tiles,
    // 4 ring slots, 74 KB: two workgroups = 8 waves per CU) ahead of 192 x 4 by 5-11 %, everything else behind
    static const int kExact[][2] = {{256, 2}, {192, 4}, {320, 2}, {128, 4}, {128, 2}, {256, 1}};
    const int (&kShapes)[6][2] = shift_wanted ? kExact : (small_tiles ? kShort : kLong);
    const int nt_env = spec_env("VND_SPEC_NT", 0);
    rr_hint = spec_env("VND_SPEC_RR", rr_hint);
    dd_hint = spec_env("VND_SPEC_DD", dd_hint);
    for (const auto &shape : kShapes) {
        SpecConfig c;
        c.bc = bc ? 1 : 0;
        c.nt = nt_env > 0 ? nt_env : shape[0];
        c.rr = rr_hint > 0 ? rr_hint : shape[1];
        c.la = spec_env("VND_SPEC_LA", c.la);
        if (c.nt % 64 != 0 || c.nt > 1024 || c.rr > 16) return false;
        const int T = c.tile();
        c.pp = (T + reach + T - 1) / T + 1;            // slots covering tile + halo, plus the one being refilled
        if (c.pp < 2) c.pp = 2;
        c.dd = std::min(dd_hint > 0 ? dd_hint : (c.pp >= 4 ? 2 : 1), 3);      // measured: 2 ahead only pays with 4+ slots
        // exact mode: shifted copies of the planes (odd offsets become aligned pairs: +22 % on class-path tables, +7 % on
        // function-path ones at cfg2) when two workgroups still fit a CU
        c.shift = (shift_wanted && c.rr >= 2 && spec_env("VND_SPEC_SHIFT", 1)) ? 1 : 0;
        if (c.shift && c.lds_bytes() > 81 * 1024) c.shift = 0;
        // ... and those two still hold six waves (128-thread workgroups measured 23 % slower with the copies than without)
        if (c.shift && ((size_t)(160 * 1024) / c.lds_bytes()) * (size_t)(c.nt / 64) < 6) c.shift = 0;
        if (c.pp <= 8 && c.lds_bytes() <= std::min<size_t>(lds_limit, (c.shift ? 81 : 64) * 1024) &&
            2 * ((size_t)c.pp * T + 2 * c.nt) * 4 < 65536) {
            *out = c;
            return true;
        }
        if (nt_env > 0 && rr_hint > 0) break;
    }
    return false;
}

inline void spec_append(std::string &s, const char *fmt, ...)
{
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    s += buf;
}

// exact float literal (hex float keeps every bit of the weight)
inline std::string spec_float(float v)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%af", (double)v);
    return buf;
}

// The generated prologue: geometry macros and the tap sequence of every channel pair.  The sequence
// round-robins over the four accumulator sets (channel 0/1 x even/odd offset) so that consecutive
// FMAs never depend on each other.
inline std::string spec_prologue(const SpecTable &t, const SpecConfig &c)
{
    const int groups = t.C / 2;
    std::string s;
    spec_append(s, "#define VS_NT %d\n#define VS_RR %d\n#define VS_PP %d\n#define VS_DD %d\n#define VS_LA %d\n", c.nt, c.rr,
                c.pp, c.dd, c.la);
    spec_append(s, "#define VS_C %d\n#define VS_GROUPS %d\n#define VS_NT_STORES %d\n#define VS_EXACT %d\n#define VS_EPI %d\n#define VS_BC %d\n#define VS_SHIFT %d\n", t.C, groups,
                c.nt_stores, c.exact, c.epi, c.bc, c.shift);
    spec_append(s, "#define VS_NT_STORE_AUX %d\n", spec_env("VND_SPEC_STORE_AUX", 2));   // cache policy bits of the non-temporal stores (tuning)
    spec_append(s, "#define VS_LOAD_AUX %d\n", spec_env("VND_SPEC_LOAD_AUX", 2));     // input is read once: non-temporal loads (+1-2 % on cfg2)
    // wider signals: the four lanes of a quad exchange their frames so that one store instruction writes four CONSECUTIVE
    // frames' pieces (one 128-byte line) instead of every other frame's (tools/micro/piece_stores.hip: 2.8 against 1.7 TB/s)
    spec_append(s, "#define VS_QUAD_STORES %d\n", spec_env("VND_SPEC_QUAD_STORES", 1));
    {   // resident workgroups per CU (LDS-bound, at most 32 waves) -> waves per SIMD the register budget must allow
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(16, 2048 / c.nt), (160 * 1024) / c.lds_bytes()));
        const int waves = (per_cu * (c.nt / 64) + 3) / 4;
        spec_append(s, "#define VS_WAVES_PER_EU %d\n", std::max(1, std::min(waves, 8)));
    }
    if (c.exact) {
        // the step program: channels alternate, each in table order; END marks a tap that closes a segment
        // (1: function-path table, the sum is the output; 2: out += segment; 3: segment *= gain first)
        struct Step { int ch, off, end; float w, gain; };
        std::vector<std::vector<Step>> prog(groups);
        size_t width = 1;
        int max_off = 0;
        for (int g = 0; g < groups; ++g) {
            std::vector<Step> per[2];
            for (int cc = 0; cc < 2; ++cc) {
                const int ch = 2 * g + cc;
                for (int32_t k = t.tap_off[ch]; k < t.tap_off[ch + 1]; ++k) per[cc].push_back(Step{cc, t.idx[k], 0, t.w_raw[k], 1.0f});
                if (per[cc].empty()) continue;
                if (!t.has_seg) {
                    per[cc].back().end = 1;
                } else {
                    for (int32_t sg = t.seg_off[ch]; sg < t.seg_off[ch + 1]; ++sg) {
                        const int32_t last = t.seg_end[sg] - 1 - t.tap_off[ch];      // spec scope: no empty segment
                        per[cc][last].end = t.apply_gain ? 3 : 2;
                        per[cc][last].gain = t.seg_gain[sg];
                    }
                }
            }
            for (size_t k = 0; k < std::max(per[0].size(), per[1].size()); ++k)
                for (int cc = 0; cc < 2; ++cc)
                    if (k < per[cc].size()) prog[g].push_back(per[cc][k]);
            width = std::max(width, prog[g].size());
            for (const Step &st : prog[g]) max_off = std::max(max_off, st.off + (st.off & 1) + 2 * c.nt * (c.rr - 1));
        }
        spec_append(s, "#define VS_MAX_OFF %d\n", max_off);
        spec_append(s, "__device__ constexpr int VS_EX_N[%d] = {", groups);
        for (int g = 0; g < groups; ++g) spec_append(s, "%zu,", prog[g].size());
        s += "};\n";
        auto emit = [&](const char *type, const char *name, int what) {
            spec_append(s, "__device__ constexpr %s %s[%d][%zu] = {", type, name, groups, width);
            for (int g = 0; g < groups; ++g) {
                s += "{";
                for (size_t k = 0; k < width; ++k) {
                    const bool in = k < prog[g].size();
                    if (what == 0) spec_append(s, "%d,", in ? prog[g][k].ch : 0);
                    else if (what == 1) spec_append(s, "%d,", in ? prog[g][k].off : 0);
                    else if (what == 2) spec_append(s, "%d,", in ? prog[g][k].end : 0);
                    else { s += spec_float(in ? (what == 3 ? prog[g][k].w : prog[g][k].gain) : 0.0f); s += ","; }
                }
                s += "},";
            }
            s += "};\n";
        };
        emit("int", "VS_EX_CH", 0);
        emit("int", "VS_EX_OFF", 1);
        emit("int", "VS_EX_END", 2);
        emit("float", "VS_EX_W", 3);
        emit("float", "VS_EX_GAIN", 4);
        s += "#define VS_DISPATCH(g) switch (g) {";
        for (int g = 0; g < groups; ++g) spec_append(s, " case %d: vs_span<%d>(a, lds, stream, span); break;", g, g);
        s += " default: break; }\n";
        return s;
    }
    struct Read { int plane, off; std::vector<int> set, row; std::vector<float> w; };
    std::vector<std::vector<Read>> sched(groups);
    std::vector<std::vector<int>> odd_off(2 * groups);
    std::vector<std::vector<float>> odd_w(2 * groups);
    size_t max_reads = 1, max_cons = 1, max_odd = 1;
    for (int g = 0; g < groups; ++g) {
        std::vector<std::pair<int, float>> sets[4];
        for (int cc = 0; cc < 2; ++cc) {
            const int ch = 2 * g + cc;
            for (int32_t k = t.tap_off[ch]; k < t.tap_off[ch + 1]; ++k) {
                if (t.w[k] == 0.0f) continue;                         // adds nothing in any summation order
                sets[cc * 2 + (t.idx[k] & 1)].push_back({t.idx[k] & ~1, t.w[k]});
            }
        }
        // taps round-robin over the four accumulator sets (consecutive FMAs never depend on each other,
        // the LDS planes alternate), rows innermost; a (plane, offset) pair already scheduled takes
        // the new consumer instead of a second read
        size_t pos[4] = {0, 0, 0, 0};
        std::vector<Read> &rd = sched[g];
        for (bool any = true; any;) {
            any = false;
            static const int order[4] = {0, 2, 1, 3};
            for (int o = 0; o < 4; ++o) {
                const int st = order[o];
                if (pos[st] >= sets[st].size()) continue;
                const int off = sets[st][pos[st]].first;
                const float w = sets[st][pos[st]].second;
                ++pos[st];
                any = true;
                for (int j = 0; j < c.rr; ++j) {
                    const int plane = c.bc ? 0 : (st >> 1), at = off + 2 * c.nt * j;      // fan-out: both channels read the one plane
                    Read *hit = nullptr;
                    for (Read &r : rd) if (r.plane == plane && r.off == at) { hit = &r; break; }
                    if (!hit) { rd.push_back(Read{plane, at, {}, {}, {}}); hit = &rd.back(); }
                    hit->set.push_back(st); hit->row.push_back(j); hit->w.push_back(w);
                }
            }
        }
        size_t cons = 0;
        for (const Read &r : rd) {
            cons += r.set.size();
            // the order in which row 0 of each channel's odd set accumulates: the span-end chain repeats it
            for (size_t m = 0; m < r.set.size(); ++m)
                if ((r.set[m] & 1) && r.row[m] == 0) { odd_off[2 * g + (r.set[m] >> 1)].push_back(r.off); odd_w[2 * g + (r.set[m] >> 1)].push_back(r.w[m]); }
        }
        max_reads = std::max(max_reads, rd.size());
        max_cons = std::max(max_cons, cons);
        for (int cc = 0; cc < 2; ++cc) max_odd = std::max(max_odd, odd_off[2 * g + cc].size());
    }
    auto open_array = [&](const char *type, const char *name, size_t width) {
        spec_append(s, "__device__ constexpr %s %s[%d][%zu] = {", type, name, groups, width);
    };
    int max_off = 0;
    for (int g = 0; g < groups; ++g) for (const Read &r : sched[g]) max_off = std::max(max_off, r.off);
    spec_append(s, "#define VS_MAX_OFF %d\n", max_off);
    spec_append(s, "__device__ constexpr int VS_RD_N[%d] = {", groups);
    for (int g = 0; g < groups; ++g) spec_append(s, "%zu,", sched[g].size());
    s += "};\n";
    open_array("int", "VS_RD_PLANE", max_reads);
    for (int g = 0; g < groups; ++g) { s += "{"; for (size_t k = 0; k < max_reads; ++k) spec_append(s, "%d,", k < sched[g].size() ? sched[g][k].plane : 0); s += "},"; }
    s += "};\n";
    open_array("int", "VS_RD_OFF", max_reads);
    for (int g = 0; g < groups; ++g) { s += "{"; for (size_t k = 0; k < max_reads; ++k) spec_append(s, "%d,", k < sched[g].size() ? sched[g][k].off : 0); s += "},"; }
    s += "};\n";
    open_array("int", "VS_RD_FIRST", max_reads + 1);
    for (int g = 0; g < groups; ++g) {
        s += "{";
        size_t run = 0;
        for (size_t k = 0; k <= max_reads; ++k) { spec_append(s, "%zu,", run); if (k < sched[g].size()) run += sched[g][k].set.size(); }
        s += "},";
    }
    s += "};\n";
    auto consumers = [&](const char *type, const char *name, int what) {
        open_array(type, name, max_cons);
        for (int g = 0; g < groups; ++g) {
            s += "{";
            size_t n = 0;
            for (const Read &r : sched[g])
                for (size_t m = 0; m < r.set.size(); ++m, ++n) {
                    if (what == 0) spec_append(s, "%d,", r.set[m]);
                    else if (what == 1) spec_append(s, "%d,", r.row[m]);
                    else { s += spec_float(r.w[m]); s += ","; }
                }
            for (; n < max_cons; ++n) s += what == 2 ? "0.0f," : "0,";
            s += "},";
        }
        s += "};\n";
    };
    consumers("int", "VS_CS_SET", 0);
    consumers("int", "VS_CS_ROW", 1);
    consumers("float", "VS_CS_W", 2);
    spec_append(s, "__device__ constexpr int VS_NODD[%d][2] = {", groups);
    for (int g = 0; g < groups; ++g) spec_append(s, "{%zu,%zu},", odd_off[2 * g].size(), odd_off[2 * g + 1].size());
    s += "};\n";
    spec_append(s, "__device__ constexpr int VS_ODD_OFF[%d][2][%zu] = {", groups, max_odd);
    for (int g = 0; g < groups; ++g) {
        s += "{";
        for (int cc = 0; cc < 2; ++cc) { s += "{"; for (size_t k = 0; k < max_odd; ++k) spec_append(s, "%d,", k < odd_off[2 * g + cc].size() ? odd_off[2 * g + cc][k] : 0); s += "},"; }
        s += "},";
    }
    s += "};\n";
    spec_append(s, "__device__ constexpr float VS_ODD_W[%d][2][%zu] = {", groups, max_odd);
    for (int g = 0; g < groups; ++g) {
        s += "{";
        for (int cc = 0; cc < 2; ++cc) { s += "{"; for (size_t k = 0; k < max_odd; ++k) { s += (k < odd_off[2 * g + cc].size() ? spec_float(odd_w[2 * g + cc][k]) : std::string("0.0f")); s += ","; } s += "},"; }
        s += "},";
    }
    s += "};\n";
    s += "#define VS_DISPATCH(g) switch (g) {";
    for (int g = 0; g < groups; ++g) spec_append(s, " case %d: vs_span<%d>(a, lds, stream, span); break;", g, g);
    s += " default: break; }\n";
    return s;
}

// kernel argument block: must match VSArgs in vnd_spec_kernel.inc
struct SpecArgs {
    const float *x;
    float *y;
    long long n;
    int tiles_total, tiles_per_span, spans;
    unsigned nblocks, units;
    int epi_ms_encode, epi_use_width;
    float epi_w_mid, epi_w_side;
    // window form only (VWArgs): a small launch's CU chunks - see vnd_win_kernel.inc
    int chunk_tiles, chunk_len0, chunks_per_stream, cus_per_xcd;
    int stagger_ticks, chunk_prio, bal_total;      // (bal_total: the balanced cut of the window form - vnd_win_kernel.inc)
    double *epi_blk_sum;          // VW_EPI with 32-frame runs: per-block sums of squares for the block-parallel exact RMS sums
    int epi_nblocks, epi_rows_major;
    unsigned *pace;               // [2048 CU indices][2] tile counters of co-resident workgroups (window form: pacing)
};

struct SpecModule {
    SpecConfig cfg;
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    bool failed = false;
    bool pending = false;        // looked for in the disk cache only (a small launch never triggers a build): not there
    bool building = false;       // a thread is compiling it outside the table's lock (spec_module)
    std::string log;
};

// ---- code-object cache on disk -----------------------------------------------------------------
// A table's kernel is a pure function of its generated source, the compile options and the hipRTC
// version: the compiled code object is kept under $VND_SPEC_CACHE_DIR (default ~/.cache/vndecorrelate_amd;
// "off" disables) under a 64-bit FNV-1a key, so only the first process that meets a table pays the
// 1.5-5 s of hipRTC.  Any I/O problem just means compiling.
inline std::string spec_cache_dir()
{
    if (const char *e = getenv("VND_SPEC_CACHE_DIR")) return std::string(e) == "off" ? std::string() : std::string(e);
    const char *home = getenv("HOME");
    return home && *home ? std::string(home) + "/.cache/vndecorrelate_amd" : std::string();
}

inline std::string spec_cache_key(const std::string &src, const char *const *opts, int nopts)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const char *p, size_t n) { for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; } };
    mix(src.data(), src.size());
    for (int i = 0; i < nopts; ++i) mix(opts[i], strlen(opts[i]) + 1);
    int major = 0, minor = 0;
    (void)hiprtcVersion(&major, &minor);
    mix((const char *)&major, sizeof major);
    mix((const char *)&minor, sizeof minor);
    char buf[32];
    snprintf(buf, sizeof buf, "%016llx", (unsigned long long)h);
    return buf;
}

// A cached object is executed as the table's kernel (in VND_MODE_EXACT: as the bit-identical one), so a file is
// trusted only behind a header that ties it to its payload and to the compiler that made it:
//   "VNDCO1\0\0" | payload bytes (u64) | FNV-1a 64 of the payload (u64) | 64 bytes of build id (hipRTC version + HIP runtime build)
// Anything else - truncated, stale after a ROCm upgrade, not ours - is a cache miss: recompile and overwrite.
struct SpecCacheHeader {
    char magic[8];
    uint64_t bytes, hash;
    char build[64];
};

inline uint64_t spec_fnv(const char *p, size_t n, uint64_t h = 1469598103934665603ull)
{
    for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
    return h;
}

inline void spec_build_id(char (&out)[64])
{
    memset(out, 0, sizeof out);
    int major = 0, minor = 0, rt = 0;
    (void)hiprtcVersion(&major, &minor);
    (void)hipRuntimeGetVersion(&rt);
    snprintf(out, sizeof out, "hiprtc %d.%d hip %d %s", major, minor, rt,
#ifdef HIP_VERSION_GITHASH
             HIP_VERSION_GITHASH
#else
             ""
#endif
    );
}

inline bool spec_cache_load(const std::string &path, std::vector<char> *code)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    bool ok = false;
    SpecCacheHeader h;
    char build[64];
    spec_build_id(build);
    if (fread(&h, 1, sizeof h, f) == sizeof h && memcmp(h.magic, "VNDCO1\0\0", 8) == 0 && h.bytes > 64 && h.bytes < ((uint64_t)1 << 28) &&
        memcmp(h.build, build, sizeof build) == 0) {
        code->resize((size_t)h.bytes);
        ok = fread(code->data(), 1, (size_t)h.bytes, f) == (size_t)h.bytes && fgetc(f) == EOF &&
             memcmp(code->data(), "\177ELF", 4) == 0 && spec_fnv(code->data(), code->size()) == h.hash;
    }
    fclose(f);
    if (!ok) code->clear();
    return ok;
}

inline void spec_cache_store(const std::string &dir, const std::string &path, const std::vector<char> &code)
{
    std::string cmd_dir = dir;
    for (size_t i = 1; i <= cmd_dir.size(); ++i)                     // mkdir -p, private to the user: the objects are executed
        if (i == cmd_dir.size() || cmd_dir[i] == '/') { const std::string part = cmd_dir.substr(0, i); (void)mkdir(part.c_str(), 0700); }
    std::string tmp = path + ".XXXXXX";                              // unique per writer (threads of one process included)
    const int fd = mkstemp(&tmp[0]);
    if (fd < 0) return;
    FILE *f = fdopen(fd, "wb");
    if (!f) { close(fd); (void)remove(tmp.c_str()); return; }
    SpecCacheHeader h;
    memcpy(h.magic, "VNDCO1\0\0", 8);
    h.bytes = code.size();
    h.hash = spec_fnv(code.data(), code.size());
    spec_build_id(h.build);
    const bool ok = fwrite(&h, 1, sizeof h, f) == sizeof h && fwrite(code.data(), 1, code.size(), f) == code.size();
    const bool closed = fclose(f) == 0;
    if (!ok || !closed || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());     // rename is atomic: readers never see half a file
}

// Bytes of private (scratch) memory per lane that a code object's kernel asks for - what a register spill shows up as - read
// from the kernel descriptor in the ELF itself (symbol <kernel>.kd: group_segment_fixed_size at byte 0, private_segment_fixed_size
// at byte 4).  hipFuncGetAttribute reports the same number, but not under every tool that wraps the runtime (under rocprofv3 it
// came back empty and a spilling build was let through); the file does not change with the observer.  -1: not found.
inline long spec_private_bytes(const std::vector<char> &code, const char *kernel)
{
    struct Ehdr { unsigned char ident[16]; uint16_t type, machine; uint32_t version; uint64_t entry, phoff, shoff; uint32_t flags;
                  uint16_t ehsize, phentsize, phnum, shentsize, shnum, shstrndx; };
    struct Shdr { uint32_t name, type; uint64_t flags, addr, offset, size; uint32_t link, info; uint64_t addralign, entsize; };
    struct Sym { uint32_t name; unsigned char info, other; uint16_t shndx; uint64_t value, size; };
    const size_t n = code.size();
    if (n < sizeof(Ehdr) || memcmp(code.data(), "\177ELF", 4) != 0 || code[4] != 2) return -1;
    Ehdr eh; memcpy(&eh, code.data(), sizeof eh);
    if (eh.shentsize != sizeof(Shdr) || eh.shoff > n || (uint64_t)eh.shnum * sizeof(Shdr) > n - eh.shoff) return -1;
    auto shdr = [&](unsigned i) { Shdr sh; memcpy(&sh, code.data() + eh.shoff + (size_t)i * sizeof(Shdr), sizeof sh); return sh; };
    const std::string want = std::string(kernel) + ".kd";
    for (unsigned i = 0; i < eh.shnum; ++i) {
        const Shdr st = shdr(i);
        if ((st.type != 2 && st.type != 11) || st.entsize != sizeof(Sym) || st.link >= eh.shnum) continue;      // SYMTAB, DYNSYM
        const Shdr str = shdr(st.link);
        if (st.offset > n || st.size > n - st.offset || str.offset > n || str.size > n - str.offset) continue;
        for (uint64_t k = 0; k + sizeof(Sym) <= st.size; k += sizeof(Sym)) {
            Sym sy; memcpy(&sy, code.data() + st.offset + k, sizeof sy);
            if (sy.name >= str.size || sy.shndx == 0 || sy.shndx >= eh.shnum) continue;
            const char *nm = code.data() + str.offset + sy.name;
            if (strnlen(nm, str.size - sy.name) != want.size() || memcmp(nm, want.data(), want.size()) != 0) continue;
            const Shdr sec = shdr(sy.shndx);
            if (sy.value < sec.addr) return -1;
            const uint64_t at = sec.offset + (sy.value - sec.addr);
            if (at > n || n - at < 8) return -1;
            uint32_t priv; memcpy(&priv, code.data() + at + 4, 4);
            return (long)priv;
        }
    }
    return -1;
}

// the WINDOW form's translation unit (vnd_win.hpp)
inline std::string win_source_for(const SpecTable &t, const SpecConfig &cfg);

inline bool spec_compile(const SpecTable &t, const SpecConfig &cfg, int device, int lds_limit, SpecModule *m, bool cache_only = false)
{
    m->cfg = cfg;
    m->pending = false;
    std::string src;
    if (cfg.win) {
        src = win_source_for(t, cfg);
    } else {
        src = spec_prologue(t, cfg);
        src += kSpecKernelSource;
    }
    if (getenv("VND_SPEC_BREAK")) src += "\n#error VND_SPEC_BREAK: injected build failure (fallback test)\n";
    if (const char *dump = getenv("VND_SPEC_DUMP")) {
        if (FILE *f = fopen(dump, "w")) { fputs(src.c_str(), f); fclose(f); }
    }
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
    std::vector<char> code;
    const std::string dir = spec_cache_dir();
    const std::string path = dir.empty() ? std::string() : dir + "/" + spec_cache_key(src, opts, 4) + ".co";
    if (path.empty() || !spec_cache_load(path, &code)) {
        if (cache_only) { m->pending = true; return false; }
        hiprtcProgram prog = nullptr;
        if (hiprtcCreateProgram(&prog, src.c_str(), "vnd_spec_kernel.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
            m->failed = true; m->log = "hiprtcCreateProgram failed";
            return false;
        }
        const hiprtcResult rc = hiprtcCompileProgram(prog, 4, opts);
        size_t log_size = 0;
        hiprtcGetProgramLogSize(prog, &log_size);
        if (log_size > 1) { m->log.resize(log_size); hiprtcGetProgramLog(prog, &m->log[0]); }
        if (rc != HIPRTC_SUCCESS) {
            hiprtcDestroyProgram(&prog);
            m->failed = true;
            return false;
        }
        size_t code_size = 0;
        hiprtcGetCodeSize(prog, &code_size);
        code.resize(code_size);
        hiprtcGetCode(prog, code.data());
        hiprtcDestroyProgram(&prog);
        if (!path.empty()) spec_cache_store(dir, path, code);
    }
    (void)device;
    if (hipModuleLoadData(&m->module, code.data()) != hipSuccess ||
        hipModuleGetFunction(&m->fn, m->module, "vnd_spec_kernel") != hipSuccess) {
        (void)hipGetLastError();
        if (!path.empty()) (void)remove(path.c_str());        // a cached object this runtime cannot load: next time, compile
        m->failed = true; m->log += " (module load failed)";
        return false;
    }
    if (cfg.win) {
        // the window form lives on its registers: a build that spills (private memory per lane) streams the spill through
        // the caches at every tile - slower than the pair-read form it was meant to beat.  Rejected; the caller falls back.
        int local = 0;
        if (hipFuncGetAttribute(&local, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, m->fn) != hipSuccess) { local = 0; (void)hipGetLastError(); }
        local = (int)std::max<long>(local, spec_private_bytes(code, "vnd_spec_kernel"));      // (the code object itself: observer-proof)
        if (local > 0 && !getenv("VND_WIN_ALLOW_SPILL")) {
            (void)hipModuleUnload(m->module);
            m->module = nullptr; m->fn = nullptr;
            m->failed = true;
            m->log += " (window form spills " + std::to_string(local) + " bytes of registers per lane: rejected)";
            if (getenv("VND_SPEC_VERBOSE")) fprintf(stderr, "vnd: window build frames_per_lane=%d threads=%d split=%d reads_ahead=%d:%s\n", cfg.win, cfg.nt, cfg.win_s, cfg.la, m->log.c_str());
            return false;
        }
    }
    if (cfg.lds_bytes() > 65536)
        (void)hipFuncSetAttribute((const void *)m->fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_limit);
    return true;
}

}  // namespace vnd
