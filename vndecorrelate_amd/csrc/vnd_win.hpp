// vnd_win.hpp - the WINDOW form of the per-table kernel: geometry and source generator (host code, pure:
// testable without a device).  The kernel's fixed part is vnd_win_kernel.inc; what is generated here is
// the prologue of geometry macros and vw_taps(), one lane's tap sum fully unrolled:
//   * the lane owns M consecutive output frames j = 0 .. M-1 of both channels;
//   * per channel the union of the taps' windows [i, i + M) is read once, in ascending 16-byte chunks
//     (elements o .. o+3 of the lane's view, o = 0 at the lane's first own frame);
//   * chunk o, half h (elements o+2h, o+2h+1) feeds, for every tap i of the channel:
//       i even:  output pair (j, j+1), j = o + 2h - i,             if 0 <= j <= M-2   (accumulator E[j/2])
//       i odd:   output pair (j, j+1), j = o + 2h - i (odd),       if 1 <= j <= M-3   (accumulator P[(j-1)/2])
//                output 0   from element o+2h+1 (j = -1),          single FMA         (accumulator O0)
//                output M-1 from element o+2h   (j = M-1),         single FMA         (accumulator OL)
//     - the reference's sum over taps of w * x[n + i] (decorrelation.py:649-658), every product exactly once.
#pragma once
#include "vnd_spec.hpp"

namespace vnd {

static const char kWinKernelSource[] =
#include "vnd_win_kernel.inc"
    ;

struct WinGeom {
    int M = 0, nt = 0, G = 8;
    int DE = 0;          // entries of halo: the farthest entry past its own that a lane reads
    int R = 0;           // ring entries
    int NB = 0;          // base registers per channel
    int plane = 0;       // bytes between chunk planes
    int npl = 2;         // input planes sets (1: mono input fanned out)
    size_t lds_bytes() const { return (size_t)npl * (size_t)(M / 4) * (size_t)plane; }
    int tile() const { return nt * M; }
};

// Geometry of the window kernel for a table; false when it does not fit (the caller keeps the pair-read kernel).
inline bool win_geometry(const SpecTable &t, int M, int nt, int G, bool bc, size_t lds_limit, WinGeom *g)
{
    if (!(M == 16 || M == 32 || M == 64) || nt % 64 != 0 || nt < 64 || nt > 1024 || G < 1) return false;
    g->M = M; g->nt = nt; g->G = G; g->npl = bc ? 1 : 2;
    g->DE = (t.max_index + M - 1) / M;
    g->R = nt + g->DE;
    g->NB = g->DE / G + 1;
    const int qc = M / 4;
    int units = g->R + G;                               // 16-byte slots of one chunk plane: ring + mirror
    // the 8-byte accesses of the staging and of the transposition touch QC planes at once: an ODD multiple of 16/QC slots
    // between planes spreads them over all banks (QC = 8: stride = 2 mod 4; QC = 4: 4 mod 8; QC = 16: odd)
    while (units % (32 / qc) != 16 / qc) ++units;
    g->plane = units * 16;
    if ((size_t)(qc - 1) * g->plane + (size_t)G * 16 >= 65536) return false;      // ds offset field
    if ((size_t)qc * g->plane >= 65536 && !bc) return false;                       // channel 1's planes as an immediate
    return g->lds_bytes() <= lds_limit;
}

struct WinOp { int kind, acc, half; float w; };          // kind 0: E pair, 1: P pair, 2: O0 (second element of the half), 3: OL (first)
struct WinRead { int ch, o; std::vector<WinOp> ops; };

// the read schedule of one channel: chunks in ascending order, each with the FMAs it feeds (taps ascending inside)
inline std::vector<WinRead> win_schedule(const SpecTable &t, int ch, int M)
{
    std::vector<std::pair<int, float>> taps;
    for (int32_t k = t.tap_off[ch]; k < t.tap_off[ch + 1]; ++k)
        if (t.w[k] != 0.0f) taps.push_back({t.idx[k], t.w[k]});
    std::stable_sort(taps.begin(), taps.end(), [](const std::pair<int, float> &a, const std::pair<int, float> &b) { return a.first < b.first; });
    std::vector<WinRead> out;
    if (taps.empty()) return out;
    const int o_last = ((taps.back().first + M - 1) / 4) * 4;
    for (int o = 0; o <= o_last; o += 4) {
        WinRead rd{ch, o, {}};
        for (const auto &tp : taps) {
            const int i = tp.first;
            for (int h = 0; h < 2; ++h) {
                const int j = o + 2 * h - i;
                if ((i & 1) == 0) {
                    if (j >= 0 && j <= M - 2) rd.ops.push_back(WinOp{0, j / 2, h, tp.second});
                } else {
                    if (j >= 1 && j <= M - 3) rd.ops.push_back(WinOp{1, (j - 1) / 2, h, tp.second});
                    else if (j == -1) rd.ops.push_back(WinOp{2, 0, h, tp.second});
                    else if (j == M - 1) rd.ops.push_back(WinOp{3, 0, h, tp.second});
                }
            }
        }
        if (!rd.ops.empty()) out.push_back(std::move(rd));
    }
    return out;
}

// LDS bytes one lane reads per tile for its tap sums, and the FMAs (tap x output) they feed: the figure of merit
inline void win_traffic(const SpecTable &t, int M, size_t *lds_bytes, size_t *fmas)
{
    *lds_bytes = 0; *fmas = 0;
    for (int ch = 0; ch < t.C; ++ch) {
        *lds_bytes += 16 * win_schedule(t, ch, M).size();
        for (int32_t k = t.tap_off[ch]; k < t.tap_off[ch + 1]; ++k) if (t.w[k] != 0.0f) *fmas += (size_t)M;
    }
}

inline std::string win_taps_function(const SpecTable &t, const WinGeom &g, int la)
{
    const int M = g.M, qc = M / 4;
    std::string s;
    s += "__device__ __forceinline__ void vw_taps(vw_lchar *const (&b)[2][VW_NB], float (&o0)[VW_M], float (&o1)[VW_M])\n{\n";
    spec_append(s, "    v4f q[%d];\n    v2f E[%d], P[%d];\n    float O0, OL;\n", la + 1, M / 2, M / 2);
    // one read stream over both channels: the pipeline stays full across the channel boundary
    std::vector<WinRead> reads;
    size_t first_of_ch[3] = {0, 0, 0};
    for (int ch = 0; ch < 2; ++ch) {
        first_of_ch[ch] = reads.size();
        for (WinRead &r : win_schedule(t, ch, M)) reads.push_back(std::move(r));
    }
    first_of_ch[2] = reads.size();
    auto emit_read = [&](size_t k) {
        const WinRead &r = reads[k];
        const int dE = r.o / M, rr = (r.o % M) / 4, kb = dE / g.G;
        spec_append(s, "    q[%zu] = VW_RD(b[%d][%d], %d);\n", k % (size_t)(la + 1), r.ch, kb, (dE - kb * g.G) * 16 + rr * g.plane);
    };
    auto emit_merge = [&](int ch, const std::vector<char> &e_used, const std::vector<char> &p_used, bool o0_used, bool ol_used) {
        for (int j = 0; j < M; ++j) {
            std::string ev = e_used[j / 2] ? ("E[" + std::to_string(j / 2) + "]." + ((j & 1) ? "y" : "x")) : std::string();
            std::string ov;
            if (j == 0) { if (o0_used) ov = "O0"; }
            else if (j == M - 1) { if (ol_used) ov = "OL"; }
            else if (p_used[(j - 1) / 2]) ov = "P[" + std::to_string((j - 1) / 2) + "]." + ((j & 1) ? "x" : "y");
            std::string rhs = ev.empty() ? (ov.empty() ? std::string("0.0f") : ov) : (ov.empty() ? ev : ev + " + " + ov);
            spec_append(s, "    o%d[%d] = %s;\n", ch, j, rhs.c_str());
        }
    };
    for (size_t k = 0; k < std::min(reads.size(), (size_t)la); ++k) emit_read(k);
    for (int ch = 0; ch < 2; ++ch) {
        std::vector<char> e_used(M / 2, 0), p_used(M / 2, 0);
        bool o0_used = false, ol_used = false;
        for (size_t k = first_of_ch[ch]; k < first_of_ch[ch + 1]; ++k) {
            if (k + la < reads.size()) emit_read(k + la);
            const std::string qk = "q[" + std::to_string(k % (size_t)(la + 1)) + "]";
            for (const WinOp &op : reads[k].ops) {
                const std::string w = spec_float(op.w);
                if (op.kind <= 1) {
                    const std::string acc = std::string(op.kind == 0 ? "E[" : "P[") + std::to_string(op.acc) + "]";
                    const std::string x = qk + (op.half ? ".zw" : ".xy");
                    char &used = op.kind == 0 ? e_used[op.acc] : p_used[op.acc];
                    if (used) spec_append(s, "    %s = VW_FMA(%s, %s, %s);\n", acc.c_str(), x.c_str(), w.c_str(), acc.c_str());
                    else spec_append(s, "    %s = VW_MUL(%s, %s);\n", acc.c_str(), x.c_str(), w.c_str());
                    used = 1;
                } else {
                    const bool is0 = op.kind == 2;
                    const std::string acc = is0 ? "O0" : "OL";
                    const std::string x = qk + (is0 ? (op.half ? ".w" : ".y") : (op.half ? ".z" : ".x"));
                    bool &used = is0 ? o0_used : ol_used;
                    if (used) spec_append(s, "    %s = __builtin_fmaf(%s, %s, %s);\n", acc.c_str(), x.c_str(), w.c_str(), acc.c_str());
                    else spec_append(s, "    %s = %s * %s;\n", acc.c_str(), x.c_str(), w.c_str());
                    used = true;
                }
            }
            s += "    VW_SB;\n";
        }
        emit_merge(ch, e_used, p_used, o0_used, ol_used);
    }
    (void)qc;
    s += "}\n";
    return s;
}

inline std::string win_prologue(const WinGeom &g, const SpecConfig &c)
{
    std::string s;
    spec_append(s, "#define VW_NT %d\n#define VW_M %d\n#define VW_R %d\n#define VW_G %d\n#define VW_NB %d\n#define VW_DE %d\n#define VW_PLANE %d\n#define VW_LA %d\n",
                g.nt, g.M, g.R, g.G, g.NB, g.DE, g.plane, c.la);
    spec_append(s, "#define VW_NT_STORES %d\n#define VW_EPI %d\n#define VW_BC %d\n#define VW_EXACT %d\n", c.nt_stores, c.epi, c.bc, c.exact);
    spec_append(s, "#define VW_NT_STORE_AUX %d\n", spec_env("VND_SPEC_STORE_AUX", 2));
    spec_append(s, "#define VW_LOAD_AUX %d\n", spec_env("VND_SPEC_LOAD_AUX", 2));
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(16, 2048 / g.nt), (160 * 1024) / g.lds_bytes()));
    const int waves = (per_cu * (g.nt / 64) + 3) / 4;
    spec_append(s, "#define VW_WAVES_PER_EU %d\n", std::max(1, std::min(waves, 8)));
    return s;
}

// the whole translation unit of the window kernel for (table, geometry)
inline std::string win_source(const SpecTable &t, const WinGeom &g, const SpecConfig &c)
{
    std::string src = win_prologue(g, c);
    const std::string fixed = kWinKernelSource;
    const std::string marker = "//@@VW_TAPS@@";
    const size_t at = fixed.find(marker);
    src += fixed.substr(0, at);
    src += win_taps_function(t, g, c.la);
    src += fixed.substr(at + marker.size());
    return src;
}

// geometry choice: the largest workgroup whose ring (tile + halo, mirror) still fits; small_tiles starts lower
// (short streams: a ring is filled once per span)
inline bool win_pick_config(const SpecTable &t, size_t lds_limit, int M, bool small_tiles, bool bc, SpecConfig *out)
{
    // the geometry that keeps the most waves on a CU (the ring is LDS-bound: tile + halo per workgroup), the larger
    // workgroup on a tie (the halo is shared by more lanes); short streams (small_tiles: a ring is filled once per
    // span) take at most 128 threads = 4096-frame tiles
    static const int kShapes[] = {256, 192, 128, 64};
    const int nt_env = spec_env("VND_SPEC_NT", 0), g_env = spec_env("VND_WIN_G", 0);
    int best_waves = 0;
    for (int k = 0; k < 4; ++k) {
        const int nt = nt_env > 0 ? nt_env : kShapes[k];
        if (small_tiles && nt_env <= 0 && nt > 128) continue;
        for (int G : {8, 4}) {
            if (g_env > 0) G = g_env;
            WinGeom g;
            if (win_geometry(t, M, nt, G, bc, lds_limit, &g)) {
                const int per_cu = (int)std::min<size_t>(std::min<size_t>(16, 2048 / nt), (160 * 1024) / g.lds_bytes());
                const int waves = per_cu * (nt / 64);
                if (waves > best_waves) {
                    best_waves = waves;
                    SpecConfig c;
                    c.nt = nt; c.win = M; c.win_g = G; c.win_lds = (int)g.lds_bytes(); c.bc = bc ? 1 : 0;
                    c.la = spec_env("VND_SPEC_LA", 6);
                    c.rr = 0; c.pp = 0; c.dd = 0;
                    *out = c;
                }
            }
            if (g_env > 0) break;
        }
        if (nt_env > 0) break;
    }
    return best_waves > 0;
}

inline std::string win_source_for(const SpecTable &t, const SpecConfig &cfg)
{
    WinGeom g;
    if (!win_geometry(t, cfg.win, cfg.nt, cfg.win_g, cfg.bc != 0, 160 * 1024, &g)) return "#error window geometry does not fit\n";
    return win_source(t, g, cfg);
}

}  // namespace vnd
