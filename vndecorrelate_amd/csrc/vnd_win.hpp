// vnd_win.hpp - the WINDOW form of the per-table kernel: geometry and source generator (host code, pure:
// testable without a device).  The kernel's fixed part is vnd_win_kernel.inc; what is generated here is
// the prologue of geometry macros and vw_taps(), one lane's tap sum fully unrolled - one such function per channel PAIR of
// the table (vw_taps, vw_taps_1, ...: a workgroup works on one pair of a span; a stereo table has the one), each over
// the pair's two LDS plane sets:
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
#include <cmath>
#include <functional>

namespace vnd {

static const char kWinKernelSource[] =
#include "vnd_win_kernel.inc"
    ;

struct WinGeom {
    int M = 0, nt = 0, G = 8;
    int C = 2;           // interleaved channels of the signal (a workgroup takes one channel PAIR)
    int DE = 0;          // entries of halo: the farthest entry past its own that a lane reads
    int R = 0;           // ring entries
    int NB = 0;          // base registers per channel
    int plane = 0;       // bytes between chunk planes
    int npl = 2;         // plane sets in LDS: one per output channel (a mono input fills only the first - the second is where channel 1's runs cross in the store phase)
    int tail = 0;        // quads / octets: entries of the window kept OUTSIDE the ring so that the ring's length R is a multiple of 16 (NH + DE = R + tail)
    int quad = 0;        // 1: a workgroup takes a channel QUAD (16 bytes of every frame), a quarter of its lanes per CHANNEL; 2: an OCTET (32 bytes), an eighth
    int split = 0;       // 1: stereo, half the workgroup's waves per CHANNEL
    int nh() const { return split ? nt / 2 : (quad ? nt / (4 * quad) : nt); }   // lanes - and ring entries of a tile - per channel pair (split, quad: per channel)
    // octets: the 4-byte staging writes and read-backs of a wave touch, per channel plane, one dword of 32 frames x 2 quads.  The frames
    // fall into banks 8a + b (a = 0..7 chunk planes an odd number of slots apart, b = 0..3 frames of a chunk): half the banks.  The second
    // quad's plane sets are shifted by 16 bytes so that it takes the other half (8a + 4 + b) instead of meeting the first in the same ones:
    // SQ_LDS_BANK_CONFLICT 22.1 M -> 14.1 M cycles per cfg5 launch (tools/ablate/RUNS.md: run_r4j) - and the launch takes the same time: the
    // staging phase is not what the LDS array limits.
    int quad_pad() const { return quad == 2 ? 16 : 0; }
    size_t lds_bytes() const { return (size_t)(quad ? 2 * quad : 1) * (size_t)npl * (size_t)(M / 4) * (size_t)plane + (size_t)quad_pad(); }
    int tile() const { return nh() * M; }
};

inline int win_workgroups_per_cu(const WinGeom &g);

// Geometry of the window kernel for a table; false when it does not fit (the caller keeps the pair-read kernel).
inline bool win_geometry(const SpecTable &t, int M, int nt, int G, bool bc, size_t lds_limit, WinGeom *g, int quad = 0, bool split = false)
{
    if (split && (t.C != 2 || quad || nt % 128 != 0)) return false;            // a stereo table, whole waves per channel
    if (!(M == 16 || M == 32 || M == 64) || nt % 64 != 0 || nt < 64 || nt > 1024 || G < 1) return false;
    // the quad / octet form: whole channel quads, whole waves per channel, a lane of a 64-frame access inside one entry
    // (4k + 2 channels ride quads too: the last quad starts at channel C - 4)
    if (quad && (quad > 2 || (t.C % (4 * quad) != 0 && !(quad == 1 && t.C % 4 == 2 && t.C >= 6)) || bc || nt % (256 * quad) != 0 || M * quad > 64 || M > 32)) return false;
    // one lane's tap sum is straight-line code, M/2 packed instructions of 8 bytes per tap: beyond ~1000 taps per channel
    // pair it would be megabytes of code for hipRTC and the 64 KB instruction cache (cfg3's 256 taps: 33 KB) - such
    // tables keep the pair-read form
    if ((int64_t)t.idx.size() * M > 32768) return false;
    if (t.C < 2 || (t.C & 1) || (bc && t.C != 2)) return false;      // whole channel pairs
    g->M = M; g->nt = nt; g->G = G; g->npl = 2; g->C = t.C; g->quad = quad; g->split = split ? 1 : 0;
    const int nh = g->nh();
    const int qc = M / 4;
    auto lay_out = [&](int de, int tail = 0) {
        g->DE = de;
        g->R = nh + g->DE - tail;
        g->tail = tail;
        g->NB = g->DE / G + 1;
        int units = g->R + G + tail;                    // 16-byte slots of one chunk plane: ring + mirror (+ tail)
        // the 8-byte accesses of the staging and of the transposition: 16 lanes at a time write (read) 32 dwords into 32 banks -
        // 8 planes x 2 halves of ONE entry with 32-frame runs (4 planes of two entries with 16, 8 of the 16 planes with 64) - so the
        // planes must lie an ODD multiple of 8/QC slots apart (QC = 8: an odd number of slots; QC = 4: 2 mod 4; QC = 16: odd).
        // Counted, not guessed (SQ_LDS_BANK_CONFLICT per launch of 512 cfg2 streams, tools/ablate/RUNS.md: run_r3f): odd 7.7 M,
        // 2 mod 4 - the rule until late round 3 - 16.2 M, 4 mod 8 56 M, 0 mod 8 137 M.
        // 32-frame runs: with the lanes' pair indices swizzled (VW_LANE_SWIZZLE, the kernel) planes 2 mod 4 slots apart make
        // BOTH the 16-lane writes and the 32-lane read-backs conflict-free
        const bool swz = qc == 8;
        const int mod_rule = swz ? 4 : std::max(2, 16 / qc), res_rule = swz ? 2 : std::max(1, 8 / qc);
        while (units % mod_rule != res_rule) ++units;
        g->plane = units * 16;
        if ((size_t)(qc - 1) * g->plane + (size_t)G * 16 >= 65536) return false;  // ds offset field
        if ((size_t)qc * g->plane >= 65536) return false;                          // channel 1's planes as an immediate
        return g->lds_bytes() <= lds_limit;
    };
    const int de = (t.max_index + M - 1) / M;
    if (!lay_out(de)) return false;
    // A wave's window read is 64 consecutive ring entries - 16 bytes each, so entry e sits in banks 4e .. 4e+3 (mod 64) - EXCEPT in
    // the wave whose lanes straddle the ring's end: there entry R-1 is followed by entry 0, and unless R is a multiple of 16
    // entries the two runs of lanes meet in the same banks (13 % of all LDS cycles on cfg2 were such conflicts: R = 301).  A few
    // entries of extra halo make the wrap invisible to the banks - taken when they cost no workgroup of residency.
    const int aligned = de + (16 - (nh + de) % 16) % 16;
    if (aligned != de) {
        const int before = win_workgroups_per_cu(*g);
        WinGeom plain = *g;
        if (!(lay_out(aligned) && win_workgroups_per_cu(*g) >= before)) {
            *g = plain;
            // quads / octets, where the longer halo does not fit (cfg5: 149 entries of eight channels are 153 of the 160 KB): the ring is
            // CUT to the multiple of 16 below, and the few entries beyond it - which only the tile's last lanes reach, with their
            // farthest taps - live in a TAIL behind the mirror, refilled a tile ahead like the ring; the lanes' reads of those entry
            // offsets go through per-lane bases of their own (ring or tail).  Same LDS, no wave straddles a ring end out of step
            // with the banks any more: SQ_LDS_BANK_CONFLICT 0.13 -> 0.0x of the LDS cycles on cfg5.
            const int cut = (nh + de) % 16;
            if (quad && spec_env("VND_WIN_TAIL", 1) != 0 && cut > 0 && nh + de - cut >= nh + G && cut <= (nt / 64) * (64 / (M * quad))) {
                if (!(lay_out(de, cut) && win_workgroups_per_cu(*g) >= before)) *g = plain;
            }
        }
    }
    return true;
}

// Workgroups a CU holds: LDS-bound, and register-bound - a lane carries its runs' accumulators, the outputs of the first
// channel and a tile of prefetched frames (about 250 VGPRs with 32-frame runs, 150 with 16, more than 256 with 64), so
// at most 2 / 3 / 1 waves per SIMD; a budget below that would spill the prefetch to scratch (measured: 3x slower).
inline int win_waves_per_simd_max(int M, bool split = false) { return split ? (M <= 32 ? 3 : 2) : (M <= 16 ? 3 : (M <= 32 ? 2 : 1)); }

inline int win_workgroups_per_cu(const WinGeom &g)
{
    const int by_lds = (int)std::min<size_t>(std::min<size_t>(16, 2048 / g.nt), (160 * 1024) / g.lds_bytes());
    const int by_regs = std::max(1, win_waves_per_simd_max(g.M, g.split != 0) * 4 / (g.nt / 64));
    return std::max(1, std::min(by_lds, by_regs));
}

struct WinOp { int kind, acc, half; float w; int tap; };  // kind 0: E pair, 1: P pair, 2: O0 (second element of the half), 3: OL (first); tap: its offset
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
                    if (j >= 0 && j <= M - 2) rd.ops.push_back(WinOp{0, j / 2, h, tp.second, i});
                } else {
                    if (j >= 1 && j <= M - 3) rd.ops.push_back(WinOp{1, (j - 1) / 2, h, tp.second, i});
                    else if (j == -1) rd.ops.push_back(WinOp{2, 0, h, tp.second, i});
                    else if (j == M - 1) rd.ops.push_back(WinOp{3, 0, h, tp.second, i});
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

// the name of channel pair pg's tap function: vw_taps for the first (a stereo table's only) pair, vw_taps_<pg> for the others
inline std::string win_taps_name(int pg) { return pg == 0 ? std::string("vw_taps") : "vw_taps_" + std::to_string(pg); }

// one CHANNEL's tap function (the split forms): vw_taps_c<ch> for a stereo table, vw_taps_<pair>c<ch> for wider ones
inline std::string win_taps_channel_name(int pg, int ch)
{
    return pg == 0 ? "vw_taps_c" + std::to_string(ch) : "vw_taps_" + std::to_string(pg) + "c" + std::to_string(ch);
}

// vw_taps_of<PG>(): the pair's function by its number, and VW_DISPATCH: the kernel's span loop instantiated per channel pair
inline std::string win_taps_dispatch(const SpecTable &t)
{
    std::string s = "template <int PG> __device__ __forceinline__ void vw_taps_of(vw_lchar *const (&b)[2][VW_NBT], float (&o0)[VW_M], float (&o1)[VW_M])\n{\n";
    for (int pg = 0; pg < t.C / 2; ++pg)
        spec_append(s, "    %sif constexpr (PG == %d) %s(b, o0, o1);\n", pg ? "else " : "", pg, win_taps_name(pg).c_str());
    s += "}\n#define VW_DISPATCH(pg) switch (pg) {";
    for (int pg = 0; pg < t.C / 2; ++pg) spec_append(s, " case %d: vw_span<%d>(a, lds, stream, t_first, ntiles, flags, pace); break;", pg, pg);
    s += " default: break; }\n";
    return s;
}

// the LDS operand of a window read: chunk (o % M) / 4 of the entry o / M past the lane's own - base register + immediate; entry offsets
// that the tile's last lanes find in the TAIL (WinGeom::tail) have a per-lane base of their own, b[ch][NB + j]
inline std::string win_rd(const WinGeom &g, int ch, int o)
{
    const int dE = o / g.M, rr = (o % g.M) / 4;
    const int first_tail = g.R - g.nh() + 1;
    char buf[96];
    if (g.tail > 0 && dE >= first_tail) snprintf(buf, sizeof buf, "VW_RD(b[%d][%d], %d)", ch, g.NB + (dE - first_tail), rr * g.plane);
    else { const int kb = dE / g.G; snprintf(buf, sizeof buf, "VW_RD(b[%d][%d], %d)", ch, kb, (dE - kb * g.G) * 16 + rr * g.plane); }
    return buf;
}

// ---- VND_MODE_FAST in the reference's CLASS-PATH association: adds inside a segment, one multiply per segment -------------
// VelvetNoise.convolve adds and subtracts x[n + i] per segment and multiplies the segment's sum by its gain once
// (decorrelation.py:402-414); every generated table has at most len(segment_envelope) distinct |w| (:621-625, :539-542).  Under the
// board's power cap a launch's time is its energy, and a v_pk_add_f32 costs less of it than a v_pk_fma_f32: the same window reads
// sustain 12-16 % more packed adds than packed FMAs per second on random data in isolation (tools/micro/lds_power.hip,
// profiles/r06_lds_power.txt) - in the launches themselves +0.5 ... +3.6 % (profiles/r06_adds_ab.txt: never slower, hence the default).
// So an accumulator holds its partial sum in UNITS of the gain g of the taps it is taking:
//     same |w| as the last tap:   acc = acc +- x                       v_pk_add_f32, the sign a neg_lo / neg_hi modifier
//     another |w| = g':           acc = fma(acc, g / g', +-x)          the one multiply per segment, riding in an FMA
//     first tap:                  acc = 0 +- x
// and the lane's finished outputs are multiplied by the last unit (one v_pk_mul_f32 per output pair).  The chunks still arrive from
// the FAR end of the window, and inside a chunk the taps now descend too, so a chain walks the segments far to near and changes its
// unit once per segment.  Every output is fl(fl(E + O) * g): a function of the table and of the output's position alone, as before.
inline bool win_adds_ok(const SpecTable &t)
{
    std::vector<float> mags;
    for (size_t k = 0; k < t.w.size(); ++k) {
        const float m = std::fabs(t.w[k]);
        if (m == 0.0f) continue;
        if (!std::isfinite(m) || m < 1e-18f || m > 1e18f) return false;         // (every ratio of two gains a normal float32)
        if (std::find(mags.begin(), mags.end(), m) == mags.end()) mags.push_back(m);
        if (mags.size() > 8) return false;
    }
    // ... and the gains must come in RUNS along the offsets, as a segmented envelope's do (decorrelation.py:621-625): every change of
    // |w| along a channel costs its chains an FMA and a rounding of the whole partial sum - a table of a few gains in scrambled order
    // (a unit change at every other tap) measured 6.2e-7 of peak from the float64 sum at 120 taps where one FMA per tap has 3.9e-7,
    // a decaying table 2.9e-7 against 2.4e-7.  At most 8 changes per channel
    for (int ch = 0; ch < t.C; ++ch) {
        std::vector<std::pair<int, float>> taps;
        for (int32_t k = t.tap_off[ch]; k < t.tap_off[ch + 1]; ++k)
            if (t.w[k] != 0.0f) taps.push_back({t.idx[k], std::fabs(t.w[k])});
        std::stable_sort(taps.begin(), taps.end(), [](const std::pair<int, float> &a, const std::pair<int, float> &b) { return a.first < b.first; });
        int changes = 0;
        for (size_t k = 1; k < taps.size(); ++k) changes += taps[k].second != taps[k - 1].second;
        if (changes > 8) return false;
    }
    return !mags.empty();
}

// the ratio that takes a sum from units of `from` to units of `to`, rounded once
inline float win_unit_ratio(float from, float to) { return (float)((double)from / (double)to); }

// one tap of one chain in that association: `acc` (a v2f pair or a single float) takes +-x of a tap of weight w; `unit` is the gain
// whose units the chain is in (0: not open yet) and becomes |w|
inline void win_emit_adds_op(std::string &s, const std::string &acc, const std::string &x, bool pair, float w, float *unit)
{
    const float mag = std::fabs(w);
    const char sign = std::signbit(w) ? '-' : '+';
    if (*unit == 0.0f) spec_append(s, "    %s = %s %c %s;\n", acc.c_str(), pair ? "Z2" : "0.0f", sign, x.c_str());
    else if (*unit == mag) spec_append(s, "    %s = %s %c %s;\n", acc.c_str(), acc.c_str(), sign, x.c_str());
    else spec_append(s, "    %s = %s(%s, %s, %s%s);\n", acc.c_str(), pair ? "VW_FMA" : "__builtin_fmaf", acc.c_str(), spec_float(win_unit_ratio(*unit, mag)).c_str(),
                     sign == '-' ? "-" : "", x.c_str());
    *unit = mag;
}

// the outputs of one channel from its chains: o[j] = E + O per output, in the chains' common unit with one packed multiply per
// output pair - or, where the chains of a channel ended in different units (a table whose nearest taps are all of one parity),
// every chain scaled by its own unit first.  name(kind, k): the chain's C expression (kind 0 E[k], 1 P[k], 2 O0, 3 OL)
template <class Name>
inline void win_emit_adds_merge(std::string &s, int ch, int M, const std::vector<float> &e_unit, const std::vector<float> &p_unit, float o0_unit, float ol_unit, Name name)
{
    float common = 0.0f;
    bool uniform = true;
    auto see = [&](float u) { if (u == 0.0f) return; if (common == 0.0f) common = u; else if (u != common) uniform = false; };
    for (float u : e_unit) see(u);
    for (float u : p_unit) see(u);
    see(o0_unit); see(ol_unit);
    for (int j = 0; j < M; ++j) {
        const float eu = e_unit[j / 2];
        const float ou = j == 0 ? o0_unit : (j == M - 1 ? ol_unit : p_unit[(j - 1) / 2]);
        std::string ev = eu != 0.0f ? (name(0, j / 2) + "." + ((j & 1) ? "y" : "x")) : std::string();
        std::string ov;
        if (ou != 0.0f) ov = j == 0 ? name(2, 0) : (j == M - 1 ? name(3, 0) : name(1, (j - 1) / 2) + "." + ((j & 1) ? "x" : "y"));
        if (!uniform) {
            if (!ev.empty()) ev = "(" + ev + " * " + spec_float(eu) + ")";
            if (!ov.empty()) ov = "(" + ov + " * " + spec_float(ou) + ")";
        }
        const std::string rhs = ev.empty() ? (ov.empty() ? std::string("0.0f") : ov) : (ov.empty() ? ev : ev + " + " + ov);
        spec_append(s, "    o%d[%d] = %s;\n", ch, j, rhs.c_str());
    }
    if (uniform && common != 0.0f && common != 1.0f) {
        const std::string u = spec_float(common);
        for (int k = 0; k < M / 2; ++k)
            spec_append(s, "    { const v2f t2 = v2f{o%d[%d], o%d[%d]} * v2f{%s, %s}; o%d[%d] = t2.x; o%d[%d] = t2.y; }\n", ch, 2 * k, ch, 2 * k + 1, u.c_str(), u.c_str(),
                        ch, 2 * k, ch, 2 * k + 1);
    }
}

inline std::string win_taps_function(const SpecTable &t, const WinGeom &g, int la, int pg = 0, int only_ch = -1, bool adds = false)
{
    const int M = g.M;
    std::string s;
    // (only_ch: the split form's per-channel function vw_taps_c<ch> - it leaves the other channel's outputs alone)
    const std::string fname = only_ch < 0 ? win_taps_name(pg) : win_taps_channel_name(pg, only_ch);
    s += "__device__ __forceinline__ void " + fname + "(vw_lchar *const (&b)[2][VW_NBT], float (&o0)[VW_M], float (&o1)[VW_M])\n{\n";
    spec_append(s, "    v4f q[%d];\n    v2f E[%d], P[%d];\n    float O0, OL;\n", la + 1, M / 2, M / 2);
    if (adds) s += "    const v2f Z2 = {0.0f, 0.0f};\n";
    // one read stream over both channels: the pipeline stays full across the channel boundary
    std::vector<WinRead> reads;
    size_t first_of_ch[3] = {0, 0, 0};
    // The chunks are consumed from the FAR end of the window to the near one: a velvet table's gains decay with the offset, so every
    // chain adds its small terms first - the partial sums stay small for most of the chain, and so does what each rounding costs
    // (128-tap tables: the worst sample's distance from the reference 8.5e-7 -> 7.9e-7 / 7.1e-7 -> 5.1e-7 of peak, 30 taps 3.6e-7 -> 3.0e-7;
    // most of what remains is the reference's own rounding).  Taps inside a chunk keep ascending order.
    const bool far_first = spec_env("VND_WIN_FAR_FIRST", 1) != 0;
    for (int ch = 0; ch < 2; ++ch) {
        first_of_ch[ch] = reads.size();
        if (only_ch >= 0 && ch != only_ch) continue;
        std::vector<WinRead> one = win_schedule(t, 2 * pg + ch, M);
        if (far_first) std::reverse(one.begin(), one.end());
        if (adds && far_first)
            for (WinRead &r : one) std::stable_sort(r.ops.begin(), r.ops.end(), [](const WinOp &a, const WinOp &b) { return a.tap > b.tap; });
        for (WinRead &r : one) { r.ch = ch; reads.push_back(std::move(r)); }      // ch: the LDS plane set
    }
    first_of_ch[2] = reads.size();
    auto emit_read = [&](size_t k) {
        const WinRead &r = reads[k];
        spec_append(s, "    q[%zu] = %s;\n", k % (size_t)(la + 1), win_rd(g, r.ch, r.o).c_str());
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
        if (only_ch >= 0 && ch != only_ch) continue;
        std::vector<char> e_used(M / 2, 0), p_used(M / 2, 0);
        bool o0_used = false, ol_used = false;
        std::vector<float> e_unit(M / 2, 0.0f), p_unit(M / 2, 0.0f);      // adds: the gain whose units the chain is in (0: not open yet)
        float o0_unit = 0.0f, ol_unit = 0.0f;
        for (size_t k = first_of_ch[ch]; k < first_of_ch[ch + 1]; ++k) {
            if (k + la < reads.size()) emit_read(k + la);
            const std::string qk = "q[" + std::to_string(k % (size_t)(la + 1)) + "]";
            for (const WinOp &op : reads[k].ops) {
                const std::string w = spec_float(op.w);
                if (adds) {
                    const bool pair = op.kind <= 1, is0 = op.kind == 2;
                    const std::string acc = pair ? std::string(op.kind == 0 ? "E[" : "P[") + std::to_string(op.acc) + "]" : std::string(is0 ? "O0" : "OL");
                    const std::string x = pair ? qk + (op.half ? ".zw" : ".xy") : qk + (is0 ? (op.half ? ".w" : ".y") : (op.half ? ".z" : ".x"));
                    win_emit_adds_op(s, acc, x, pair, op.w, pair ? (op.kind == 0 ? &e_unit[op.acc] : &p_unit[op.acc]) : (is0 ? &o0_unit : &ol_unit));
                    continue;
                }
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
        if (adds)
            win_emit_adds_merge(s, ch, M, e_unit, p_unit, o0_unit, ol_unit, [](int kind, int k) {
                return kind == 0 ? "E[" + std::to_string(k) + "]" : (kind == 1 ? "P[" + std::to_string(k) + "]" : std::string(kind == 2 ? "O0" : "OL"));
            });
        else emit_merge(ch, e_used, p_used, o0_used, ol_used);
    }
    s += "}\n";
    return s;
}

// ---- VND_MODE_EXACT in the window form ----------------------------------------------------------------
// The reference's association, bit for bit: per output ONE accumulator that takes the taps in TABLE order,
//     sb = f32(sb + f32(x * w))        (NumPy's  out[:N-i] += x[i:] * w,  decorrelation.py:656-658;
//                                       a weight of +-1 is the class path's  sb += / -= x,  :405-410)
// and at a segment's end  sb *= gain  (unless the envelope is the identity, :411-412),  out += sb  (:413).
// A lane's chunks arrive in ascending order, so for every output the taps of an ascending run of the table
// arrive in table order: a PASS is a maximal run of consecutive taps (within one segment) with non-decreasing
// offsets - the whole table on the function path, the negative and the positive list of each segment on the
// class path (:247-254) - and reads the union of ITS taps' windows.  Within a pass:
//   * the accumulators are the aligned output pairs (0,1), (2,3), ... and EVERY update is one v_pk_add_f32.  Output
//     pair (j, j+1) takes from tap i the input pair starting at element e = i + j: for even e an aligned half of a
//     chunk (A0 = elements o, o+1; A1 = o+2, o+3), for odd e a pair formed once per chunk from its registers
//     (B1 = o+1, o+2; B0 = o-1, o with the previous chunk's last element: one v_pk_mov_b32 each) and shared by all
//     the odd taps that meet the chunk.  A B0 pair is met one chunk late; every later tap of the same output still
//     comes after it, so the order per output stays the table's;
//   * f32(x * w) is formed once per (chunk, pair, |w|) and shared by the taps of that magnitude; f32(x * -g) = -f32(x * g)
//     and sb + (-p) = sb - p exactly, so the sign rides in the add;
//   * accumulators open with 0 + v / 0 - v (NumPy starts from zeros: 0 + -0 = +0), never with a bare copy.
struct WinExTap { int idx; float w; };
struct WinExPass { int ch; std::vector<WinExTap> taps; int end; float gain; };      // end: 0 (the segment goes on), 1 sum is the output, 2 out += sb, 3 sb *= gain first
struct WinExOp { int type, k; float w; };                  // type 0 A0, 1 A1, 2 B0, 3 B1; accumulator pair k
struct WinExRead { int ch, o; size_t pass; std::vector<WinExOp> ops; };

inline std::vector<WinExPass> win_exact_passes(const SpecTable &t, int ch)
{
    std::vector<WinExPass> out;
    const int32_t k0 = t.tap_off[ch], k1 = t.tap_off[ch + 1];
    if (k0 == k1) return out;
    auto add_segment = [&](int32_t a, int32_t b, int end, float gain) {
        int32_t start = a;
        for (int32_t k = a + 1; k <= b; ++k) {
            if (k == b || t.idx[k] < t.idx[k - 1]) {
                WinExPass ps{ch, {}, k == b ? end : 0, gain};
                for (int32_t m = start; m < k; ++m) ps.taps.push_back(WinExTap{t.idx[m], t.w_raw[m]});
                out.push_back(std::move(ps));
                start = k;
            }
        }
    };
    if (!t.has_seg) {
        add_segment(k0, k1, 1, 1.0f);
    } else {
        int32_t prev = k0;
        for (int32_t sg = t.seg_off[ch]; sg < t.seg_off[ch + 1]; ++sg) {
            add_segment(prev, t.seg_end[sg], t.apply_gain ? 3 : 2, t.seg_gain[sg]);      // spec scope: no empty segment
            prev = t.seg_end[sg];
        }
    }
    return out;
}

// the chunks a pass reads, ascending, each with its updates in table order (and the chunks that only lend
// their last element to the next one's B0 pair)
inline void win_exact_reads(const std::vector<WinExPass> &passes, int M, std::vector<WinExRead> *reads, size_t pass_base)
{
    for (size_t p = 0; p < passes.size(); ++p) {
        const WinExPass &ps = passes[p];
        std::map<int, std::vector<WinExOp>> by_chunk;
        const bool singles = true;                          // (an odd offset as single adds: the packed form with shuffled pairs only tied with the pair-read kernel)
        for (const WinExTap &tp : ps.taps) {
            if (singles && (tp.idx & 1)) {
                // an odd offset: output j takes element idx + j.  The aligned input pairs (types 0 / 1) then feed ONE element
                // each of two neighbouring accumulator pairs - single adds (types 4..7: pair type | 4, k = the OUTPUT index of
                // the pair's first element, -1 / M - 1 at the run's edges where only one element has an output)
                for (int j = -1; j < M; j += 2) {
                    const int e = tp.idx + j;                           // even: an aligned pair (e, e + 1) -> outputs (j, j + 1)
                    by_chunk[(e & 3) == 0 ? e : e - 2].push_back(WinExOp{((e & 3) == 0 ? 0 : 1) | 4, j, tp.w});
                }
                continue;
            }
            for (int j = 0; j < M; j += 2) {
                const int e = tp.idx + j;
                switch (e & 3) {
                case 0: by_chunk[e].push_back(WinExOp{0, j / 2, tp.w}); break;
                case 2: by_chunk[e - 2].push_back(WinExOp{1, j / 2, tp.w}); break;
                case 1: by_chunk[e - 1].push_back(WinExOp{3, j / 2, tp.w}); break;
                default: by_chunk[e + 1].push_back(WinExOp{2, j / 2, tp.w}); break;
                }
            }
        }
        std::vector<int> lend;
        for (const auto &kv : by_chunk)
            for (const WinExOp &op : kv.second)
                if (op.type == 2) { lend.push_back(kv.first - 4); break; }
        for (int o : lend) by_chunk[o];                          // present, possibly without updates
        for (auto &kv : by_chunk) reads->push_back(WinExRead{ps.ch, kv.first, pass_base + p, std::move(kv.second)});
    }
}

inline void win_traffic_exact(const SpecTable &t, int M, size_t *lds_bytes, size_t *fmas)
{
    *lds_bytes = 0; *fmas = 0;
    for (int ch = 0; ch < t.C; ++ch) {
        std::vector<WinExRead> reads;
        const std::vector<WinExPass> passes = win_exact_passes(t, ch);
        if (!passes.empty()) win_exact_reads(passes, M, &reads, 0);
        *lds_bytes += 16 * reads.size();
        *fmas += (size_t)M * (size_t)(t.tap_off[ch + 1] - t.tap_off[ch]);
    }
}

// A mono input fanned out (VW_BC, plain form): both output channels read the SAME plane, and the two channels of a velvet table
// place their taps almost alike (the same segment grid, jittered) - so ONE read stream over the union of both channels' windows feeds
// both channels' FMAs: cfg2's table 178 reads per tile and lane instead of 157 + 160, 1.48 B of LDS per FMA instead of 2.64.  Each
// channel keeps its own E / P chains in the read order of win_taps_function - FAR end of the window first by default (small terms first,
// VND_WIN_FAR_FIRST), ascending offsets when that is switched off: the results are those of a pass per channel, bit for bit.
inline std::string win_taps_function_merged(const SpecTable &t, const WinGeom &g, int la, int pg = 0, bool adds = false)
{
    const int M = g.M;
    std::string s;
    s += "__device__ __forceinline__ void " + win_taps_name(pg) + "(vw_lchar *const (&b)[2][VW_NBT], float (&o0)[VW_M], float (&o1)[VW_M])\n{\n";
    spec_append(s, "    v4f q[%d];\n    v2f E0[%d], P0[%d], E1[%d], P1[%d];\n    float O00, OL0, O01, OL1;\n", la + 1, M / 2, M / 2, M / 2, M / 2);
    if (adds) s += "    const v2f Z2 = {0.0f, 0.0f};\n";
    struct Rd { int o; std::vector<std::pair<int, WinOp>> ops; };           // (channel, op)
    std::map<int, Rd> by_o;
    for (int ch = 0; ch < 2; ++ch)
        for (const WinRead &r : win_schedule(t, 2 * pg + ch, M)) {
            Rd &rd = by_o[r.o];
            rd.o = r.o;
            for (const WinOp &op : r.ops) rd.ops.push_back({ch, op});
        }
    std::vector<Rd> reads;
    for (auto &kv : by_o) reads.push_back(std::move(kv.second));
    if (spec_env("VND_WIN_FAR_FIRST", 1) != 0) {      // (small terms first: win_taps_function)
        std::reverse(reads.begin(), reads.end());
        if (adds)
            for (Rd &r : reads) std::stable_sort(r.ops.begin(), r.ops.end(), [](const std::pair<int, WinOp> &a, const std::pair<int, WinOp> &b) { return a.second.tap > b.second.tap; });
    }
    auto emit_read = [&](size_t k) {
        spec_append(s, "    q[%zu] = %s;\n", k % (size_t)(la + 1), win_rd(g, 0, reads[k].o).c_str());
    };
    std::vector<float> e_unit[2] = {std::vector<float>(M / 2, 0.0f), std::vector<float>(M / 2, 0.0f)}, p_unit[2] = {std::vector<float>(M / 2, 0.0f), std::vector<float>(M / 2, 0.0f)};
    float o0_unit[2] = {0.0f, 0.0f}, ol_unit[2] = {0.0f, 0.0f};      // adds: the chains' units (win_taps_function)
    std::vector<char> e_used[2] = {std::vector<char>(M / 2, 0), std::vector<char>(M / 2, 0)}, p_used[2] = {std::vector<char>(M / 2, 0), std::vector<char>(M / 2, 0)};
    bool o0_used[2] = {false, false}, ol_used[2] = {false, false};
    for (size_t k = 0; k < std::min(reads.size(), (size_t)la); ++k) emit_read(k);
    for (size_t k = 0; k < reads.size(); ++k) {
        if (k + la < reads.size()) emit_read(k + la);
        const std::string qk = "q[" + std::to_string(k % (size_t)(la + 1)) + "]";
        for (const auto &cop : reads[k].ops) {
            const int ch = cop.first;
            const WinOp &op = cop.second;
            const std::string w = spec_float(op.w), c = std::to_string(ch);
            if (adds) {
                const bool pair = op.kind <= 1, is0 = op.kind == 2;
                const std::string acc = pair ? std::string(op.kind == 0 ? "E" : "P") + c + "[" + std::to_string(op.acc) + "]" : std::string(is0 ? "O0" : "OL") + c;
                const std::string x = pair ? qk + (op.half ? ".zw" : ".xy") : qk + (is0 ? (op.half ? ".w" : ".y") : (op.half ? ".z" : ".x"));
                win_emit_adds_op(s, acc, x, pair, op.w, pair ? (op.kind == 0 ? &e_unit[ch][op.acc] : &p_unit[ch][op.acc]) : (is0 ? &o0_unit[ch] : &ol_unit[ch]));
                continue;
            }
            if (op.kind <= 1) {
                const std::string acc = std::string(op.kind == 0 ? "E" : "P") + c + "[" + std::to_string(op.acc) + "]";
                const std::string x = qk + (op.half ? ".zw" : ".xy");
                char &used = op.kind == 0 ? e_used[ch][op.acc] : p_used[ch][op.acc];
                if (used) spec_append(s, "    %s = VW_FMA(%s, %s, %s);\n", acc.c_str(), x.c_str(), w.c_str(), acc.c_str());
                else spec_append(s, "    %s = VW_MUL(%s, %s);\n", acc.c_str(), x.c_str(), w.c_str());
                used = 1;
            } else {
                const bool is0 = op.kind == 2;
                const std::string acc = std::string(is0 ? "O0" : "OL") + c;
                const std::string x = qk + (is0 ? (op.half ? ".w" : ".y") : (op.half ? ".z" : ".x"));
                bool &used = is0 ? o0_used[ch] : ol_used[ch];
                if (used) spec_append(s, "    %s = __builtin_fmaf(%s, %s, %s);\n", acc.c_str(), x.c_str(), w.c_str(), acc.c_str());
                else spec_append(s, "    %s = %s * %s;\n", acc.c_str(), x.c_str(), w.c_str());
                used = true;
            }
        }
        s += "    VW_SB;\n";
    }
    for (int ch = 0; ch < 2; ++ch) {
        const std::string c = std::to_string(ch);
        if (adds) {
            win_emit_adds_merge(s, ch, M, e_unit[ch], p_unit[ch], o0_unit[ch], ol_unit[ch], [&](int kind, int k) {
                return kind == 0 ? "E" + c + "[" + std::to_string(k) + "]" : (kind == 1 ? "P" + c + "[" + std::to_string(k) + "]" : (kind == 2 ? "O0" : "OL") + c);
            });
            continue;
        }
        for (int j = 0; j < M; ++j) {
            std::string ev = e_used[ch][j / 2] ? ("E" + c + "[" + std::to_string(j / 2) + "]." + ((j & 1) ? "y" : "x")) : std::string();
            std::string ov;
            if (j == 0) { if (o0_used[ch]) ov = "O0" + c; }
            else if (j == M - 1) { if (ol_used[ch]) ov = "OL" + c; }
            else if (p_used[ch][(j - 1) / 2]) ov = "P" + c + "[" + std::to_string((j - 1) / 2) + "]." + ((j & 1) ? "x" : "y");
            const std::string rhs = ev.empty() ? (ov.empty() ? std::string("0.0f") : ov) : (ov.empty() ? ev : ev + " + " + ov);
            spec_append(s, "    o%d[%d] = %s;\n", ch, j, rhs.c_str());
        }
    }
    s += "}\n";
    return s;
}

inline std::string win_taps_function_exact(const SpecTable &t, const WinGeom &g, int la, int pg = 0, int only_ch = -1)
{
    const int M = g.M;
    const size_t ring = (size_t)la + 2;                  // read k lands in q[k % ring]: the previous chunk stays whole while read k + la is issued
    std::string s;
    const std::string fname = only_ch < 0 ? win_taps_name(pg) : win_taps_channel_name(pg, only_ch);
    s += "__device__ __forceinline__ void " + fname + "(vw_lchar *const (&b)[2][VW_NBT], float (&o0)[VW_M], float (&o1)[VW_M])\n{\n";
    spec_append(s, "    v4f q[%zu];\n    v2f S[%d], A[%d];\n    const v2f Z2 = {0.0f, 0.0f};\n", ring, M / 2, M / 2);
    std::vector<WinExPass> passes;
    std::vector<WinExRead> reads;
    size_t pass_first[3] = {0, 0, 0};
    for (int ch = 0; ch < 2; ++ch) {
        pass_first[ch] = passes.size();
        if (only_ch >= 0 && ch != only_ch) continue;
        std::vector<WinExPass> ps = win_exact_passes(t, 2 * pg + ch);
        for (WinExPass &x : ps) x.ch = ch;                           // the LDS plane set
        if (!ps.empty()) win_exact_reads(ps, M, &reads, passes.size());
        for (WinExPass &x : ps) passes.push_back(std::move(x));
    }
    pass_first[2] = passes.size();
    auto emit_read = [&](size_t k) {
        const WinExRead &r = reads[k];                                   // (a lent chunk may lie before the lane's run only if o < 0: never, o >= 0)
        spec_append(s, "    q[%zu] = %s;\n", k % ring, win_rd(g, r.ch, r.o).c_str());
    };
    for (size_t k = 0; k < std::min(reads.size(), (size_t)la); ++k) emit_read(k);
    size_t rk = 0;
    for (int ch = 0; ch < 2; ++ch) {
        if (only_ch >= 0 && ch != only_ch) continue;
        std::vector<char> e_live(M, 0);          // per OUTPUT: the segment accumulator holds a value
        bool a_live = false;                     // the channel's output accumulators hold a value (class path)
        bool sum_is_output = false;
        for (size_t p = pass_first[ch]; p < pass_first[ch + 1]; ++p) {
            const WinExPass &ps = passes[p];
            for (; rk < reads.size() && reads[rk].pass == p; ++rk) {
                if (rk + la < reads.size()) emit_read(rk + la);
                const WinExRead &rd = reads[rk];
                if (rd.ops.empty()) continue;                   // lends its last element to the next chunk only
                const std::string qk = "q[" + std::to_string(rk % ring) + "]";
                s += "    {\n";
                bool need_b0 = false, need_b1 = false;
                for (const WinExOp &op : rd.ops) { need_b0 |= op.type == 2; need_b1 |= op.type == 3; }
                if (need_b0) spec_append(s, "        const v2f x2 = vw_pair(q[%zu].zw, %s.xy);\n", (rk + ring - 1) % ring, qk.c_str());
                if (need_b1) spec_append(s, "        const v2f x3 = vw_pair(%s.xy, %s.zw);\n", qk.c_str(), qk.c_str());
                const std::string xs[4] = {qk + ".xy", qk + ".zw", "x2", "x3"};
                // the products this chunk needs: one per (|w| != 1, pair)
                // (all the products first, then the sums: a sum right behind its product waits for it)
                std::vector<std::pair<float, int>> prods;
                std::string sums;
                auto value_of = [&](float w, int type) -> std::string {
                    const float m = std::fabs(w);
                    if (m == 1.0f) return xs[type];
                    size_t at = 0;
                    while (at < prods.size() && !(prods[at].first == m && prods[at].second == type)) ++at;
                    if (at == prods.size()) {
                        prods.push_back({m, type});
                        const std::string gw = spec_float(m);
                        spec_append(s, "        const v2f p%zu = %s * v2f{%s, %s};\n", at, xs[type].c_str(), gw.c_str(), gw.c_str());
                    }
                    return "p" + std::to_string(at);
                };
                for (const WinExOp &op : rd.ops) {
                    const char sign = std::signbit(op.w) ? '-' : '+';
                    if (op.type & 4) {
                        // single sums: element 0 of the pair -> output k, element 1 -> output k + 1 (where they exist)
                        const std::string v = value_of(op.w, op.type & 3);
                        for (int e = 0; e < 2; ++e) {
                            const int j = op.k + e;
                            if (j < 0 || j >= M) continue;
                            const std::string acc = "S[" + std::to_string(j / 2) + "]." + ((j & 1) ? "y" : "x");
                            const std::string first = e_live[j] ? acc : std::string("0.0f");
                            spec_append(sums, "        %s = %s(%s, %s.%s);\n", acc.c_str(), sign == '-' ? "vw_sub1" : "vw_add1", first.c_str(), v.c_str(), e ? "y" : "x");
                            e_live[j] = 1;
                        }
                        continue;
                    }
                    const std::string v = value_of(op.w, op.type);
                    const int j0 = 2 * op.k;
                    if (e_live[j0] && e_live[j0 + 1]) {
                        spec_append(sums, "        S[%d] = S[%d] %c %s;\n", op.k, op.k, sign, v.c_str());
                    } else if (!e_live[j0] && !e_live[j0 + 1]) {
                        spec_append(sums, "        S[%d] = Z2 %c %s;\n", op.k, sign, v.c_str());
                    } else {
                        if (!e_live[j0]) spec_append(sums, "        S[%d].x = 0.0f;\n", op.k);
                        if (!e_live[j0 + 1]) spec_append(sums, "        S[%d].y = 0.0f;\n", op.k);
                        spec_append(sums, "        S[%d] = S[%d] %c %s;\n", op.k, op.k, sign, v.c_str());
                    }
                    e_live[j0] = e_live[j0 + 1] = 1;
                }
                s += sums;
                s += "    }\n    VW_SB;\n";
            }
            if (ps.end == 1) {
                sum_is_output = true;
            } else if (ps.end >= 2) {
                const std::string gw = spec_float(ps.gain);
                for (int k = 0; k < M / 2; ++k) {
                    if (ps.end == 3) spec_append(s, "    S[%d] = S[%d] * v2f{%s, %s};\n", k, k, gw.c_str(), gw.c_str());
                    if (a_live) spec_append(s, "    A[%d] = A[%d] + S[%d];\n", k, k, k);
                    else spec_append(s, "    A[%d] = Z2 + S[%d];\n", k, k);
                }
                a_live = true;
                std::fill(e_live.begin(), e_live.end(), 0);
            }
        }
        for (int j = 0; j < M; ++j) {
            const char *src = sum_is_output ? "S" : "A";
            const bool have = sum_is_output ? (e_live[j] != 0) : a_live;
            if (have) spec_append(s, "    o%d[%d] = %s[%d].%s;\n", ch, j, src, j / 2, (j & 1) ? "y" : "x");
            else spec_append(s, "    o%d[%d] = 0.0f;\n", ch, j);
        }
    }
    s += "}\n";
    return s;
}

// VND_MODE_EXACT, a mono input fanned out through a FUNCTION-path stereo table (VW_BC, plain form): both channels read the one plane
// set, so ONE ascending read stream over the union of both channels' windows feeds both channels' sums - each channel still takes its
// taps in table order (the stream ascends, and so does a function-path table: one pass per channel, win_exact_passes), every sum is the
// same float32 operation on the same operands as in a pass per channel: bit-identical.  And f32(x * |w|) is formed once per (chunk,
// pair, |w|) for BOTH channels: a velvet table's two channels share their segment grid, the nearest taps often their very offsets.
inline bool win_exact_merged_ok(const SpecTable &t)
{
    if (t.C != 2 || t.has_seg) return false;
    for (int ch = 0; ch < 2; ++ch) if (win_exact_passes(t, ch).size() != 1) return false;      // ascending: one pass per channel
    return true;
}

inline std::string win_taps_function_exact_merged(const SpecTable &t, const WinGeom &g, int la, int pg = 0)
{
    const int M = g.M;
    const size_t ring = (size_t)la + 2;
    std::string s;
    s += "__device__ __forceinline__ void " + win_taps_name(pg) + "(vw_lchar *const (&b)[2][VW_NBT], float (&o0)[VW_M], float (&o1)[VW_M])\n{\n";
    spec_append(s, "    v4f q[%zu];\n    v2f S0[%d], S1[%d];\n    const v2f Z2 = {0.0f, 0.0f};\n", ring, M / 2, M / 2);
    struct Rd { int o; std::vector<std::pair<int, WinExOp>> ops; };           // (channel, update), each channel's in table order
    std::map<int, Rd> by_o;
    for (int ch = 0; ch < 2; ++ch) {
        std::vector<WinExRead> one;
        const std::vector<WinExPass> ps = win_exact_passes(t, 2 * pg + ch);
        if (!ps.empty()) win_exact_reads(ps, M, &one, 0);
        for (const WinExRead &r : one) {
            Rd &rd = by_o[r.o];
            rd.o = r.o;
            for (const WinExOp &op : r.ops) rd.ops.push_back({ch, op});
        }
    }
    std::vector<Rd> reads;
    for (auto &kv : by_o) reads.push_back(std::move(kv.second));
    auto emit_read = [&](size_t k) { spec_append(s, "    q[%zu] = %s;\n", k % ring, win_rd(g, 0, reads[k].o).c_str()); };
    for (size_t k = 0; k < std::min(reads.size(), (size_t)la); ++k) emit_read(k);
    std::vector<char> live[2] = {std::vector<char>(M, 0), std::vector<char>(M, 0)};
    for (size_t rk = 0; rk < reads.size(); ++rk) {
        if (rk + la < reads.size()) emit_read(rk + la);
        const Rd &rd = reads[rk];
        if (rd.ops.empty()) continue;
        const std::string qk = "q[" + std::to_string(rk % ring) + "]";
        s += "    {\n";
        const std::string xs[2] = {qk + ".xy", qk + ".zw"};
        std::vector<std::pair<float, int>> prods;
        std::string sums;
        auto value_of = [&](float w, int type) -> std::string {
            const float m = std::fabs(w);
            if (m == 1.0f) return xs[type];
            size_t at = 0;
            while (at < prods.size() && !(prods[at].first == m && prods[at].second == type)) ++at;
            if (at == prods.size()) {
                prods.push_back({m, type});
                const std::string gw = spec_float(m);
                spec_append(s, "        const v2f p%zu = %s * v2f{%s, %s};\n", at, xs[type].c_str(), gw.c_str(), gw.c_str());
            }
            return "p" + std::to_string(at);
        };
        for (const auto &cop : rd.ops) {
            const int ch = cop.first;
            const WinExOp &op = cop.second;
            const std::string S = ch ? "S1" : "S0";
            std::vector<char> &e_live = live[ch];
            const char sign = std::signbit(op.w) ? '-' : '+';
            const std::string v = value_of(op.w, op.type & 1);       // (aligned pairs only: odd offsets are single sums - types 4 | pair)
            if (op.type & 4) {
                for (int e = 0; e < 2; ++e) {
                    const int j = op.k + e;
                    if (j < 0 || j >= M) continue;
                    const std::string acc = S + "[" + std::to_string(j / 2) + "]." + ((j & 1) ? "y" : "x");
                    const std::string first = e_live[j] ? acc : std::string("0.0f");
                    spec_append(sums, "        %s = %s(%s, %s.%s);\n", acc.c_str(), sign == '-' ? "vw_sub1" : "vw_add1", first.c_str(), v.c_str(), e ? "y" : "x");
                    e_live[j] = 1;
                }
                continue;
            }
            const int j0 = 2 * op.k;
            if (!e_live[j0] && !e_live[j0 + 1]) {
                spec_append(sums, "        %s[%d] = Z2 %c %s;\n", S.c_str(), op.k, sign, v.c_str());
            } else {
                if (!e_live[j0]) spec_append(sums, "        %s[%d].x = 0.0f;\n", S.c_str(), op.k);
                if (!e_live[j0 + 1]) spec_append(sums, "        %s[%d].y = 0.0f;\n", S.c_str(), op.k);
                spec_append(sums, "        %s[%d] = %s[%d] %c %s;\n", S.c_str(), op.k, S.c_str(), op.k, sign, v.c_str());
            }
            e_live[j0] = e_live[j0 + 1] = 1;
        }
        s += sums;
        s += "    }\n    VW_SB;\n";
    }
    for (int ch = 0; ch < 2; ++ch)
        for (int j = 0; j < M; ++j) {
            if (live[ch][j]) spec_append(s, "    o%d[%d] = S%d[%d].%s;\n", ch, j, ch, j / 2, (j & 1) ? "y" : "x");
            else spec_append(s, "    o%d[%d] = 0.0f;\n", ch, j);
        }
    s += "}\n";
    return s;
}

inline std::string win_prologue(const WinGeom &g, const SpecConfig &c, bool two_sums = false)
{
    std::string s;
    spec_append(s, "#define VW_NT %d\n#define VW_M %d\n#define VW_R %d\n#define VW_G %d\n#define VW_NB %d\n#define VW_DE %d\n#define VW_PLANE %d\n#define VW_LA %d\n",
                g.nt, g.M, g.R, g.G, g.NB, g.DE, g.plane, c.la);
    spec_append(s, "#define VW_NT_STORES %d\n#define VW_EPI %d\n#define VW_BC %d\n#define VW_EXACT %d\n#define VW_C %d\n", c.nt_stores, c.epi, c.bc, c.exact, g.C);
    spec_append(s, "#define VW_Q %d\n#define VW_S %d\n#define VW_QUADPAD %d\n#define VW_TAIL %d\n", g.quad, g.split, g.quad_pad(), g.tail);
    // split form: how many of a wave's M/4 refill accesses per tile are loaded late (at the start of the store phase that consumes
    // them) instead of a tile ahead: 64-frame runs keep half of them out of the tap phase's registers
    // (the fast mode's E / P accumulators are twice the exact mode's sums: all but one late there - hipRTC's build of cfg2's table
    //  spills 20-64 bytes with 12-14 of the 16 late and none with 15; 4 to 15 late run the same)
    // (two_sums: a class-path table in the exact mode carries a segment sum AND an output sum per pair - as many registers as the fast mode)
    spec_append(s, "#define VW_LATE %d\n", g.split ? std::min(std::max(spec_env("VND_WIN_SPLIT_LATE", g.M >= 64 ? ((c.exact && !two_sums) ? g.M / 8 : g.M / 4 - 1) : 0), 0), g.M / 4 - 1) : 0);
    spec_append(s, "#define VW_NT_STORE_AUX %d\n", spec_env("VND_SPEC_STORE_AUX", 2));
    // the transposition as interleaved frame pairs (one 16-byte read-back per store, planes an odd number of slots apart) or as
    // planar chunks read back in 8-byte halves (VND_WIN_XPOSE_PAIRS=0: then 32-frame runs swizzle their lanes' pair indices)
    const int xpose = c.win_xpose ? 1 : 0;
    spec_append(s, "#define VW_XPOSE_PAIRS %d\n", xpose);
    spec_append(s, "#define VW_LANE_SWIZZLE %d\n", g.M == 32 ? 1 : 0);
    spec_append(s, "#define VW_STAMP_PHASES %d\n", spec_env("VND_WIN_STAMP_PHASES", 1) != 0 ? 1 : 0);
    spec_append(s, "#define VW_STAMP_WAVE %d\n", std::max(0, spec_env("VND_WIN_STAMP_WAVE", 0)));      // (whose clock readings the phase stamps are)
    spec_append(s, "#define VW_STAMPS %d\n", std::min(std::max(spec_env("VND_WIN_STAMPS", 0), 0), 4096));
    // s_setprio of the store / refill phase (0: none): cfg2 +1.0 % fast, +0.5 % exact at 1, 2 or 3; cfg3 unchanged (tools/closed/win_phase_try.py)
    spec_append(s, "#define VW_PRIO %d\n", spec_env("VND_WIN_PRIO", 1));
    spec_append(s, "#define VW_LOAD_AUX %d\n", spec_env("VND_SPEC_LOAD_AUX", 2));
    const int waves = (win_workgroups_per_cu(g) * (g.nt / 64) + 3) / 4;
    spec_append(s, "#define VW_WAVES_PER_EU %d\n", std::max(1, std::min(waves, win_waves_per_simd_max(g.M, g.split != 0))));
    return s;
}

// the whole translation unit of the window kernel for (table, geometry)
inline std::string win_source(const SpecTable &t, const WinGeom &g, const SpecConfig &c)
{
    std::string src = win_prologue(g, c, c.exact && t.has_seg);
    const std::string fixed = kWinKernelSource;
    const std::string marker = "//@@VW_TAPS@@";
    const size_t at = fixed.find(marker);
    src += fixed.substr(0, at);
    if (g.quad) {
        // quads / octets: one function per channel, the workgroup's 4Q of them picked by the wave's channel number
        const int nch = 4 * g.quad;
        for (int pg = 0; pg < t.C / 2; ++pg)
            for (int ch = 0; ch < 2; ++ch) src += c.exact ? win_taps_function_exact(t, g, c.la, pg, ch) : win_taps_function(t, g, c.la, pg, ch, c.adds != 0);
        src += "template <int PG> __device__ void vw_taps_of(vw_lchar *const (&b)[2][VW_NBT], float (&o0)[VW_M], float (&o1)[VW_M]);      // (vw_span, vw_span_q: not instantiated)\n";
        src += "template <int QD> __device__ __forceinline__ void vw_taps_of_channel(int pc, vw_lchar *const (&b)[2][VW_NBT], float (&o)[VW_M])\n{\n";
        const int nq = (t.C + nch - 1) / nch;                          // (4k + 2 channels, quads: the last one starts at channel C - 4)
        for (int qd = 0; qd < nq; ++qd) {
            const int pair0 = (t.C % nch != 0 && qd == nq - 1) ? (t.C - nch) / 2 : qd * nch / 2;
            spec_append(src, "    %sif constexpr (QD == %d) {\n        switch (pc) {\n", qd ? "else " : "", qd);
            for (int pc = 0; pc < nch; ++pc)
                spec_append(src, "        %s %s(b, o, o); break;\n", pc + 1 < nch ? ("case " + std::to_string(pc) + ":").c_str() : "default:",
                            win_taps_channel_name(pair0 + pc / 2, pc & 1).c_str());
            src += "        }\n    }\n";
        }
        src += "}\n#define VW_TAPS_OF_CHANNEL(pc) vw_taps_of_channel<QD>(pc, b, o);\n#define VW_DISPATCH(pg) switch (pg) {";
        for (int qd = 0; qd < nq; ++qd) spec_append(src, " case %d: vw_span_qc<%d>(a, lds, stream, t_first, ntiles, flags, pace); break;", qd, qd);
        src += " default: break; }\n";
    } else if (g.split) {
        for (int ch = 0; ch < 2; ++ch) src += c.exact ? win_taps_function_exact(t, g, c.la, 0, ch) : win_taps_function(t, g, c.la, 0, ch, c.adds != 0);
        src += "template <int PG> __device__ void vw_taps_of(vw_lchar *const (&b)[2][VW_NBT], float (&o0)[VW_M], float (&o1)[VW_M]);      // (vw_span: not instantiated)\n";
        src += "#define VW_DISPATCH(pg) vw_span_s(a, lds, stream, t_first, ntiles, flags, pace);\n";
    } else {
        const bool merged = c.bc && !c.exact;      // (one read stream for both channels of a mono input)
        const bool merged_exact = c.bc && c.exact && win_exact_merged_ok(t) && spec_env("VND_WIN_EXACT_MERGED", 1) != 0;
        for (int pg = 0; pg < t.C / 2; ++pg) src += c.exact ? (merged_exact ? win_taps_function_exact_merged(t, g, c.la, pg) : win_taps_function_exact(t, g, c.la, pg)) : (merged ? win_taps_function_merged(t, g, c.la, pg, c.adds != 0) : win_taps_function(t, g, c.la, pg, -1, c.adds != 0));
        src += win_taps_dispatch(t);
    }
    src += fixed.substr(at + marker.size());
    return src;
}

// geometry choice: the largest workgroup whose ring (tile + halo, mirror) still fits; small_tiles starts lower
// (short streams: a ring is filled once per span)
inline bool win_pick_config(const SpecTable &t, size_t lds_limit, int M, bool small_tiles, bool bc, SpecConfig *out,
                            const std::function<bool(const SpecConfig &)> &rejected = nullptr, int quad = 0, bool split = false,
                            bool exact = false)
{
    // the geometry that keeps the most waves on a CU (the ring is LDS-bound: tile + halo per workgroup), the larger
    // workgroup on a tie (the halo is shared by more lanes); short streams (small_tiles: a ring is filled once per
    // span) take at most 128 threads = 4096-frame tiles
    static const int kShapes[] = {512, 384, 256, 192, 128, 64};
    const int nt_env = spec_env("VND_SPEC_NT", 0), g_env = spec_env("VND_WIN_G", 0);
    int best_waves = 0;
    for (int k = 0; k < 6; ++k) {
        const int nt = nt_env > 0 ? nt_env : kShapes[k];
        if (nt_env <= 0 && nt > 256 && quad < 2 && !split) continue;  // 512 lanes: octets (an eighth of them per channel) and the split form only
        if (nt_env <= 0 && nt == 384) continue;                       // (six waves land unevenly on four SIMDs: the split form 17 % slower than with 256 lanes)
        if (small_tiles && nt_env <= 0 && nt > (quad ? 256 * quad : (split ? spec_env("VND_WIN_SPLIT_SMALL_NT", 256) : 128))) continue;
        for (int G : {8, 4}) {
            if (g_env > 0) G = g_env;
            WinGeom g;
            if (win_geometry(t, M, nt, G, bc, lds_limit, &g, quad, split)) {
                const int waves = win_workgroups_per_cu(g) * (nt / 64);
                if (waves > best_waves) {
                    SpecConfig c;
                    c.nt = nt; c.win = M; c.win_g = G; c.win_lds = (int)g.lds_bytes(); c.bc = bc ? 1 : 0; c.win_q = g.quad; c.win_s = g.split;
                    c.win_per_cu = win_workgroups_per_cu(g);
                    // reads kept in flight: each holds 4 registers, and 32-frame runs already live at ~240 of the 256 a lane
                    // has at two waves per SIMD (measured: 3 to 10 reads ahead run the same, tools/closed/win_try.py)
                    c.la = spec_env("VND_SPEC_LA", (split && M >= 64) ? (exact ? 3 : 2) : (M >= 32 ? 4 : 6));      // (64-frame runs: 64 / 128 accumulator registers)
                    c.rr = 0; c.pp = 0; c.dd = 0;
                    // the store phase: interleaved frame pairs (one 16-byte read-back per store) unless that build spilled
                    // before - it holds both channels' outputs interleaved - then planar chunks in 8-byte halves
                    c.win_xpose = (split || quad) ? 0 : (spec_env("VND_WIN_XPOSE_PAIRS", 1) != 0 ? 1 : 0);      // (the split form's outputs cross waves as planar runs)
                    if (rejected && rejected(c)) {                  // a build of this geometry failed or spilled before
                        if (!c.win_xpose) {
                            // (the split form with 64-frame runs lives within a few registers of its budget: one read less in flight
                            //  frees four - a mono input's build spills 12-20 bytes with two reads ahead and none with one)
                            bool found = false;
                            if (split && M >= 64 && spec_env("VND_SPEC_LA", -1) < 0)
                                while (!found && c.la > 1) { c.la -= 1; found = !rejected(c); }
                            if (!found) continue;
                            best_waves = waves;
                            *out = c;
                            continue;
                        }
                        // (cfg2's fast kernel: 44 bytes of spill with 4 reads ahead, none with 3 - and 3 to 10 run the same)
                        bool found = false;
                        if (c.la > 3 && spec_env("VND_SPEC_LA", -1) < 0) {
                            c.la -= 1;
                            found = !rejected(c);
                            if (!found) c.la += 1;
                        }
                        if (!found) {
                            c.win_xpose = 0;
                            if (rejected(c)) continue;
                        }
                    }
                    best_waves = waves;
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
    if (!win_geometry(t, cfg.win, cfg.nt, cfg.win_g, cfg.bc != 0, 160 * 1024, &g, cfg.win_q, cfg.win_s != 0)) return "#error window geometry does not fit\n";
    // the launch passes cfg.win_lds - computed when the plan was made - as the dynamic LDS size: a geometry that comes out different
    // here (a tuning variable changed between plan and build) must be a rejected build, never a kernel that indexes past its LDS
    if ((int)g.lds_bytes() != cfg.win_lds) return "#error window geometry changed between the launch plan and the build\n";
    return win_source(t, g, cfg);
}

}  // namespace vnd
