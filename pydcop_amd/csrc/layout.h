// layout.h -- host-side compilation of a flat factor graph (include/maxsum_gpu.h)
// into the device layout the gfx950 kernels sweep.  Pure C++, no HIP.
//
// Device layout (T = f64 or f32 element) -- the "gather-only" layout chosen from
// the measurements in profiles/r01_pattern_bench_v1.jsonl (tools/pattern_bench.hip):
//
//   F2V[2]   factor->variable messages, FACTOR-major (internal edge order): the
//            factor side writes its outputs as one coalesced stream of full
//            lines and finds the message it sent last cycle (damping and the
//            send filter, pydcop/algorithms/maxsum.py:346-377) in the old buffer
//            of the same array; the variable side gathers its inputs from it.
//   V2F[2]   variable->factor messages, VARIABLE-major (slot order of the
//            variable classes): written coalesced by the variable side, which
//            also reads its own previous output from the old buffer
//            (maxsum.py:537-564); gathered by the factor side.
//            Two buffers each: a cycle reads only cycle t-1 (Jacobi; SURVEY.md
//            Appendix A).  Per edge and cycle every side moves 3 messages -- the
//            algorithmic minimum -- no store is partial or scattered, and the only
//            irregular accesses are two message-sized gathers.
//   message  D elements padded to H (8, 16 or a multiple of 32 bytes), so a
//            gathered message never straddles a 32-byte sector.
//   tables   per factor class; uniform classes are stored entry-major (SoA:
//            entry k of factor j at k*n+j) so that a wave reads them fully
//            coalesced; generic factors keep row-major tables.
//   counters send counters (`_prev_messages[..][1]`, maxsum.py:303,474):
//            cF factor-major, cV in the variable classes' slot order -- each is
//            private to the side that owns it, so neither is ever gathered.
//   variables internal order = sorted by class then degree.
//   factors  internal order = sorted by class, then (sort_factors) by the internal
//            position of their first scope variable; the F2V records of a binary
//            register class are split by scope position (all position-0 records, then
//            all position-1 records).
//
// Factors and variables are grouped into classes; one 256-thread block works on
// one class, and one launch sweeps all classes of both sides.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/maxsum_gpu.h"

namespace mxs {

// kinds of work a block can do -------------------------------------------------
enum Kind : int32_t {
    K_F_UNARY = 1,  // arity 1, D in {2,3,4}: thread per factor, registers
    K_F_BIN = 2,    // arity 2, D x D, D in {2,3,4}: thread per factor, registers
    K_F_GEN = 3,    // anything: thread per edge, scalar loops
    K_F_NARY = 4,   // arity 2..5 with 64..1024 entries per value of the first variable (arity 3..5: any number up to
                    // 1024 once the table has 64 entries): workgroup per factor, wavefront min-reductions (own launch)
    K_V_PACK = 5,   // D in {2,3,4}, 1 <= deg <= 64: one lane per incoming edge, the
                    // variables of a wave have the same degree and are packed side
                    // by side (64/deg per wave), cross-lane sums
    K_V_GEN = 6,    // anything: thread per variable, scalar loops
    K_V_WIDE = 7,   // 9 <= D <= 256 (5..8: K_V_PACK8) or deg > 64, deg * D <= 1024: a workgroup per run of variables
                    // of one domain size (WideBlock), messages staged in LDS (own launch)
    K_V_PACK8 = 8,  // 5 <= D <= 8, 1 <= deg <= 64: the lane-per-edge scheme of K_V_PACK on records of 8 elements,
                    // the variable's own D at run time (own launch, k_variable_pack8)
    K_V_HUB = 9,    // everything the classes above leave (round 6): deg > 64 with D <= 8, deg > 256, deg * D > 1024, D > 256 --
                    // the hub variables of a scale-free graph.  One WORKGROUP per 256 outgoing edges of ONE variable (HubBlock), a lane
                    // per edge walking the reference's serial chains over the other edges; rides in the sweep launch, first in
                    // its grid (kernels.h variable_hub)
};

#ifndef MXS_BLOCK
#define MXS_BLOCK 256  // threads per workgroup of the sweep (other values: experiments only)
#endif
constexpr int BLOCK = MXS_BLOCK;

#ifndef MXS_FACTORS_SECOND_DEFAULT
#define MXS_FACTORS_SECOND_DEFAULT 0  // layout_flags bit9 forces it on, bit10 off
#endif
#ifndef MXS_SORT_FACTORS_DEFAULT
#define MXS_SORT_FACTORS_DEFAULT 1  // layout_flags bit7 forces it on, bit8 off
#endif
#ifndef MXS_SCHEDULE_DEFAULT
#define MXS_SCHEDULE_DEFAULT 1  // layout_flags bit11 (2048) forces the block schedule off, bit12 (4096) on
#endif
#ifndef MXS_TILE_BYTES
#define MXS_TILE_BYTES (512 << 10)  // window of the tiled factor order (layout.cpp); flags bit17 (131072) on, bit18 (262144) off
#endif
#ifndef MXS_TILE_RESIDENT_BYTES
#define MXS_TILE_RESIDENT_BYTES (256ll << 20)  // 8-byte words: tiled only when a cycle's bytes fit the Infinity Cache
#endif
#ifndef MXS_COMPACT_TABLES_DEFAULT
#define MXS_COMPACT_TABLES_DEFAULT 1  // layout_flags bit13 (8192) forces full-width tables, bit14 (16384) compact
#endif
// Storage type of a register class's cost tables (ClassInfo::tab_type).  Cost tables are read-only
// and, in practice, mostly small integers (colouring penalties, meeting utilities): a class
// whose every entry is exactly representable in a narrower type is stored in it and widened on
// load -- lossless, and 4.5x (i8) fewer table bytes per cycle than f64.
enum TabType : int32_t { TAB_FULL = 0, TAB_F32 = 1, TAB_I16 = 2, TAB_I8 = 3 };
constexpr int tab_elem_bytes(int t) { return t == TAB_I8 ? 1 : t == TAB_I16 ? 2 : t == TAB_F32 ? 4 : 0; }
// bytes of one factor's compact record: its entries back to back, padded so that a lane reads
// it with whole dword / dwordx2 / dwordx4 loads and a wave reads consecutive records
constexpr int tab_record_bytes(int entries, int elem) {
    const int b = entries * elem;
    return b <= 4 ? 4 : b <= 8 ? 8 : (b + 15) / 16 * 16;
}
// Workgroup-per-factor tables in a narrow type are stored LANE-PACKED: for every d0 and every
// lane of the factor's block one slot of 4 / 8 / 16 bytes holding the lane's NJ entries
// (q = lane + j * NT) back to back -- one aligned vector load per (d0, lane) brings what NJ
// full-width loads brought, so that several d0-batches can be in flight at once.
constexpr int nary_slot_bytes(int nj, int elem) {
    const int b = nj * elem;
    return b <= 4 ? 4 : b <= 8 ? 8 : 16;
}
// position of table entry (d0, q) in the lane-packed image of a factor (bytes from its start)
constexpr int64_t nary_packed_pos(int64_t d0, int64_t q, int nt, int slot, int elem) {
    return (d0 * nt + q % nt) * slot + (q / nt) * elem;
}
// ---- box layout of a narrow arity-3 table (kernels: nary_box.h, k_factor_box3) --------------------------
// ONE WAVE per factor; lane (l0, l1, l2) of an L0 x L1 x L2 grid (L0 * L1 * L2 = 64) owns the sub-box
// [l0*B0, (l0+1)*B0) x [l1*B1, ..) x [l2*B2, ..) of the table: its E = B0*B1*B2 entries back to back
// (i0 slowest) are the lane's RECORD, words = ceil(E * elem / 4) dwords that the lane keeps in registers.
// In memory the records are interleaved in 16-byte pieces -- piece k of lane l at (k * 64 + l) * 16, the
// last (words % 4) dwords of all lanes behind them -- so that every load instruction of the wave reads
// one contiguous run.  Every output's running minima then live in registers (B0 + B1 + B2 of them per lane)
// and no minimum is reduced across lanes before the factor's last entry has been read.
constexpr int BOX_SHAPES[][3] = {{2, 2, 2}, {3, 3, 3}, {4, 4, 4}, {6, 6, 6}};  // shape id = index + 1
constexpr int BOX_N_SHAPES = 4;
constexpr int BOX_WAVES = 4;        // factors (waves) per workgroup
constexpr int BOX_MAX_SUMD = 128;   // D0 + D1 + D2: two element passes of a wave in the epilogue
constexpr int BOX_MAX_WORDS = 64;   // dwords of a lane's record that are in registers at a time
// A record of up to twice that is worked through in TWO passes over the leading box digit (round 5: int16 tables on the 6 x 6 x 6
// shape -- 108 dwords per lane): it must be whole 16-byte pieces and the box's leading extent even.
constexpr bool box_record_fits(int b0, int b1, int b2, int elem) {
    const int words = (b0 * b1 * b2 * elem + 3) / 4;
    return words <= BOX_MAX_WORDS || (words <= 2 * BOX_MAX_WORDS - 8 && words % 4 == 0 && b0 % 2 == 0);
}
constexpr int box_rec_words(int entries, int elem) { return (entries * elem + 3) / 4; }
constexpr bool box_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
// shape id (1 ..) of the box layout an arity-3 table of `elem`-byte entries can take, 0 = none
// lanes along a dimension of d values cut into boxes of b: ceil(d / b) rounded up to a power of two.  Round 5: the
// dimensions need not be multiples of the box -- the lane grid may overhang the table (24 x 23 x 22 on 4 x 4 x 4
// lanes of 6 x 6 x 6 boxes); the digits past a domain are staged as +inf, the semiring's identity, and their
// (zero-filled) entries never win a minimum.
constexpr int box_lanes(int d, int b) {
    int l = 1;
    while (l * b < d) l *= 2;
    return l;
}
#ifndef MXS_BOX_MAX_PAD_PCT
#define MXS_BOX_MAX_PAD_PCT 240  // padded entries (= instructions) at most this many percent of the table's: the box kernel at 10
#endif                           // instructions per padded entry beats the lane-packed workgroup kernel even on 18^3 tables in a 24^3 grid
                                 // (meeting_50k_hetero 338.8 -> 246.2 us, f32 233.1 -> 161.3 with 240 against 150: profiles/r05_box_pad_cap_ab_v1.txt)
constexpr int nary_box_shape(int d0, int d1, int d2, int elem) {
    int best = 0;
    long best_cells = 0;
    for (int s = 0; s < BOX_N_SHAPES; ++s) {
        const int b0 = BOX_SHAPES[s][0], b1 = BOX_SHAPES[s][1], b2 = BOX_SHAPES[s][2];
        const int l0 = box_lanes(d0, b0), l1 = box_lanes(d1, b1), l2 = box_lanes(d2, b2);
        if (l0 * l1 * l2 != 64) continue;
        if (l0 > 16 || l1 > 16 || l2 > 16) continue;  // >= 4 lanes share every digit (16-byte LDS reads)
        if (l0 * b0 + l1 * b1 + l2 * b2 > BOX_MAX_SUMD) continue;
        if (!box_record_fits(b0, b1, b2, elem)) continue;
        const long cells = 64L * b0 * b1 * b2;
        if (cells * 100 > (long)MXS_BOX_MAX_PAD_PCT * d0 * d1 * d2) continue;
        if (!best || cells < best_cells) {
            best = s + 1;
            best_cells = cells;
        }
    }
    return best;
}
// ---- lane-grid layout of a BINARY (or unary) factor's table (kernels: bin_box.h, k_factor_bin) ------------
// For the factors the register classes cannot take (a domain of more than MAX_REG_D values, or two different
// ones) and the workgroup-per-factor kernel neither (fewer than 64 entries per value of the first variable):
// what the reference's own generators emit for meeting scheduling (meetingscheduling.py:450-454, 588-599:
// binary tables over 18..24 slots) and for colourings with 5..8 colours (graphcoloring.py:271).
// G = L0 x L1 lanes work on one factor, 64 / G factors share a wave.  Lane (l0, l1) owns the rows
// d0 = l0 + r * L0 (r < B0) and the columns d1 = l1 * B1 + i1 (i1 < B1): B0 * B1 table entries in registers,
// B0 partial minima towards variable 0 and B1 towards variable 1, merged across the group's lanes ONCE per
// factor through LDS.  The image is the row-major table with every row cut into L1 lane pieces of
// roundup(B1 * elem, 4) bytes -- D0 rows, none padded: a lane whose row does not exist re-reads the last one and
// the +inf staged for a digit past the domain (the semiring's identity) keeps the entry out of every minimum.
// Stored in ctables for EVERY storage type, full width included (`elem` = the word).
struct Bin2Shape {
    int L0, L1, B0, B1;
};
constexpr Bin2Shape BIN2_SHAPES[] = {
    {2, 2, 2, 2},  {2, 2, 3, 3},  {2, 2, 4, 4},                                  // 4 lanes:  up to 4x4, 6x6, 8x8
    {4, 4, 3, 3},  {4, 4, 4, 4},  {4, 4, 5, 5},  {4, 4, 6, 6},  {4, 4, 8, 8},    // 16 lanes: up to 12, 16, 20, 24, 32
    {8, 8, 5, 5},  {8, 8, 6, 6},  {8, 8, 8, 8},                                  // 64 lanes: up to 40, 48, 64
    {16, 1, 1, 1}, {16, 1, 2, 1}, {16, 1, 4, 1},                                 // unary factors: up to 16, 32, 64 values
};
constexpr int BIN2_N_SHAPES = (int)(sizeof(BIN2_SHAPES) / sizeof(BIN2_SHAPES[0]));
constexpr int BIN2_BASE = 32;      // NaryLaunch::box >= BIN2_BASE: shape BIN2_SHAPES[box - BIN2_BASE]
constexpr int BIN2_WAVES = 4;      // waves per workgroup
constexpr int BIN2_MAX_D = 64;
// ---- small-domain layout of an n-ary table (kernels: small_box.h, k_factor_small) -------------------------------
// Arity 3..5, every domain at most SMALL_P values, a narrow storage type: G = small_group(arity) lanes per factor; the
// leading small_lead(arity) digits of an entry pick the lane (index in radix SMALL_P), the trailing ones its place in the
// lane's RECORD of SMALL_P^trailing entries (radix SMALL_P, last digit fastest; digits past a domain: zero-filled).
// The image is the records of the lanes in use back to back.
constexpr int SMALL_BASE = 64;     // NaryLaunch::box == SMALL_BASE: the small-domain kernel
constexpr int SMALL_P = 5;
constexpr int SMALL_WAVES = 4;     // waves per workgroup
constexpr bool is_small(int box) { return box >= SMALL_BASE; }
constexpr int small_lead(int arity) { return arity == 3 ? 1 : 2; }
constexpr int small_group(int arity) { return arity == 3 ? 8 : 32; }
constexpr int small_pow(int e) {
    int r = 1;
    for (int i = 0; i < e; ++i) r *= SMALL_P;
    return r;
}
constexpr int small_rec_bytes(int arity, int elem) { return (small_pow(arity - small_lead(arity)) * elem + 3) / 4 * 4; }
constexpr bool is_bin2(int box) { return box >= BIN2_BASE && box < SMALL_BASE; }
constexpr int bin2_piece_bytes(int b1, int elem) { return (b1 * elem + 3) / 4 * 4; }
// the shape (index + BIN2_BASE) a table of d0 x d1 entries takes (d1 = 1, unary = true: a unary factor), 0 = none:
// the fewest image bytes + message slots, then the fewest lanes
constexpr int bin2_shape_for(int d0, int d1, bool unary) {
    int best = 0;
    long best_cost = 0;
    for (int s = 0; s < BIN2_N_SHAPES; ++s) {
        const Bin2Shape sh = BIN2_SHAPES[s];
        if ((sh.L1 == 1) != unary) continue;
        if (sh.L0 * sh.B0 < d0 || sh.L1 * sh.B1 < d1) continue;
        const long cost = (long)(sh.L0 * sh.B0) * (sh.L1 * sh.B1) * 64 + sh.L0 * sh.L1;  // padded entries, then lanes
        if (!best || cost < best_cost) {
            best = s + BIN2_BASE;
            best_cost = cost;
        }
    }
    return best;
}
// NaryLaunch::nj of the multi-pass groups of the full-width workgroup kernel (4 entries per lane and pass, BLOCK lanes)
constexpr int NARY_NJ_MULTI = 15;
constexpr int NARY_MULTI_MAX_ARITY = 6;
constexpr int64_t NARY_MULTI_MAX_R = 65536;   // the lanes' digit arithmetic divides by multiply-high: exact well beyond this
// Where entry k (row-major) of a workgroup-per-factor table lives in its narrow image.
struct NaryPlace {
    int32_t box;          // 0: lane-packed slots (nary_packed_pos), else the box shape id (>= BIN2_BASE: lane grid of a binary table)
    int32_t elem;         // bytes per entry
    int32_t nt, slot;     // lane-packed: threads of the block, bytes of a slot
    int32_t R;            // lane-packed: entries per value of the first variable
    int32_t d1, d2;       // box: domain sizes of dimensions 1 and 2
    int32_t arity;        // small-domain layout: the scope
    int32_t dom[6];
    int32_t multi;        // box == 0: a multi-pass group -- the narrow image is the row-major table itself
};
constexpr int64_t nary_place_pos(const NaryPlace& p, int64_t k) {
    if (p.box == 0) return p.multi ? k * p.elem : nary_packed_pos(k / p.R, k % p.R, p.nt, p.slot, p.elem);
    if (is_small(p.box)) {
        int x[6] = {0, 0, 0, 0, 0, 0};
        int64_t rem = k;
        for (int i = p.arity - 1; i >= 0; --i) {
            x[i] = (int)(rem % p.dom[i]);
            rem /= p.dom[i];
        }
        const int L = small_lead(p.arity);
        int64_t lane = 0, e = 0;
        for (int i = 0; i < L; ++i) lane = lane * SMALL_P + x[i];
        for (int i = L; i < p.arity; ++i) e = e * SMALL_P + x[i];
        return lane * small_rec_bytes(p.arity, p.elem) + e * p.elem;
    }
    if (is_bin2(p.box)) {
        const Bin2Shape sh = BIN2_SHAPES[p.box - BIN2_BASE];
        const int64_t piece = bin2_piece_bytes(sh.B1, p.elem), x1 = k % p.d1, x0 = k / p.d1;
        return x0 * (sh.L1 * piece) + (x1 / sh.B1) * piece + (x1 % sh.B1) * p.elem;
    }
    const int b0 = BOX_SHAPES[p.box - 1][0], b1 = BOX_SHAPES[p.box - 1][1], b2 = BOX_SHAPES[p.box - 1][2];
    const int64_t x2 = k % p.d2, x1 = (k / p.d2) % p.d1, x0 = k / ((int64_t)p.d1 * p.d2);
    const int l1n = box_lanes(p.d1, b1), l2n = box_lanes(p.d2, b2);
    const int64_t lane = ((x0 / b0) * l1n + x1 / b1) * l2n + x2 / b2;
    const int64_t byte = (((x0 % b0) * b1 + x1 % b1) * b2 + x2 % b2) * p.elem;
    const int words = box_rec_words(b0 * b1 * b2, p.elem), full = words / 4, rest = words % 4;
    const int64_t piece = byte / 16;
    return piece < full ? (piece * 64 + lane) * 16 + byte % 16
                        : (int64_t)full * 1024 + lane * rest * 4 + byte % 16;
}
// bytes of the narrow image of one factor (D0 = its first domain size)
constexpr int64_t nary_place_bytes(const NaryPlace& p, int D0) {
    if (p.box == 0) return p.multi ? ((int64_t)D0 * p.R * p.elem + 15) / 16 * 16 + 16 : (int64_t)D0 * p.nt * p.slot;  // (+ 16: a lane's run may end behind the last row)
    if (is_small(p.box)) return ((int64_t)small_pow(small_lead(p.arity)) * small_rec_bytes(p.arity, p.elem) + 15) / 16 * 16;
    if (is_bin2(p.box)) {
        const Bin2Shape sh = BIN2_SHAPES[p.box - BIN2_BASE];
        return ((int64_t)D0 * sh.L1 * bin2_piece_bytes(sh.B1, p.elem) + 15) / 16 * 16;
    }
    const int* b = BOX_SHAPES[p.box - 1];
    return (int64_t)64 * 4 * box_rec_words(b[0] * b[1] * b[2], p.elem);
}
constexpr int NUM_XCD = 8;  // MI355X: workgroup b of a grid runs on XCD b % 8 (observed; used for speed only)
constexpr int MAX_REG_D = 4;
// The reference has no limit on the arity of a constraint (maxsum.py:411-421 walks any scope); a table of
// more than 2^31 entries cannot be addressed here, so 30 binary variables is the most a factor can have.
constexpr int MAX_ARITY = MXS_MAX_ARITY;
constexpr int MAX_PACK_DEG = 64;  // one wave
constexpr int MAX_PACK8_D = 8;    // K_V_PACK8: domains of 5..8 values, records of 8 elements
constexpr int MAX_CLASSES = 24;  // block_base table travels in the kernel arguments

// Padded message length (elements) for a domain of D values of `word` bytes.
// Records of register-class size (<= 32 bytes) are NOT padded: D = 3 is 24 (f64) / 12 (f32) bytes, not
// 32 / 16 -- a quarter fewer message bytes per cycle; the send counters, which rode in the padding,
// live in cF / cV.  Measured (profiles/r03_tight_records_ab_v1.txt): coloring_100k 19.8 -> 18.7 us
// (f32 14.4 -> 13.6), the 1M instance 255 -> 242.  MXS_TIGHT=0 builds the round-2 layout.
#ifndef MXS_TIGHT
#define MXS_TIGHT 1
#endif
constexpr int half_stride(int D, int word) {
    const int bytes = D * word;
    if (MXS_TIGHT && bytes <= 32 && D <= 4) return D;  // (5..8 values: 8 elements in both widths -- K_V_PACK8's record)
    const int padded = bytes <= 8 ? 8 : bytes <= 16 ? 16 : (bytes + 31) / 32 * 32;
    return padded / word;
}

struct ClassInfo {       // one per class, read with one scalar load
    int32_t kind;
    int32_t D;           // uniform domain size (0 for generic classes)
    int32_t H;           // padded message length of the class (uniform classes)
    int32_t wait_halo;   // 1: a cut factor class of a shard -- in the fused sharded launch its
                         // blocks first wait for the halo exchange of the previous cycle
    int32_t first;       // first internal factor / variable id of the class
    int32_t count;       // number of factors / variables (K_F_GEN: edges; K_V_PACK: lanes)
    int32_t edge_base;   // first internal edge id (factor classes)
    int32_t start_only;  // K_V_GEN class of degree-0 variables: only cycle 0
    int64_t f2v_base;    // factor classes: element offset of the class in F2V
    int64_t tab_base;    // element offset of the class's tables
    int64_t cost_base;   // element offset of the class's variable costs
    int32_t block_base;  // index of the class's first block in its launch
    int32_t per_block;   // items a block covers
    int64_t ell_base;    // K_V_PACK: first lane of the class in the per-lane tables
    int64_t cv_base;     // K_V_PACK: first send counter of the class in cV
    int64_t v2f_base;    // K_V_PACK: element offset of the class in V2F
    int64_t f2v_base1;   // K_F_BIN: element offset of the class's position-1 records in F2V
                         // (the records of a binary class are split by scope position)
    int32_t tab_type;    // TabType: how the class's tables are stored
    int32_t ctab_rec;    // compact types: bytes per factor record in ctables
    int64_t ctab_base;   // compact types: byte offset of the class's records in ctables
    int32_t own_pos;     // K_F_BIN of a shard's cut factors: 1 / 2 = only the message to scope position 0 / 1 is computed
                         // (the other variable is a ghost: nobody reads what this replica would send it); 0 = both
    int32_t uni_D;       // K_V_PACK8: the domain size every variable of the class has, 0 when they differ (then the kernel
                         // reads vdom / vcost_off per variable: one more dependent load)
};

constexpr int NARY_DESC_ARITY = 6;  // slots of a descriptor; the workgroup-per-factor kernels exist up to arity 5 (round 6: the
                                    // reference's `generate secp --max_model_size 4` writes model constraints of arity 5)
struct NaryDesc {  // one per workgroup-per-factor (K_F_NARY) factor: everything its block
                   // needs, read with ONE scalar load (no chain of dependent loads)
    int64_t tab_off;     // element offset of its row-major table in the full-width image -- or, in a
                         // launch group with a narrow NaryLaunch::tab_type, BYTE offset in ctables
    int32_t edge_base;   // internal id of its first edge
    int32_t arity;
    int32_t dom[NARY_DESC_ARITY];      // domain sizes in dimensions order (1 beyond the arity)
    int32_t v2f_off[NARY_DESC_ARITY];  // V2F offsets of the incoming messages
    int32_t f2v_off[NARY_DESC_ARITY];  // F2V offsets of the outgoing messages
    uint32_t magic[NARY_DESC_ARITY];   // ceil(2^32 / dom[i]) (0 for dom[i] = 1): x / dom[i] == (x * magic[i]) >> 32 for the
                                       // x < 2^16 the kernel divides (a lane's index into the table rows)
};

// One workgroup of the K_V_WIDE launch: a run of consecutive variables of the class with the SAME
// domain size D, cut so that the staged F->V elements (slots * D), the outgoing edges (slots)
// and the own costs (variables * D) fit the kernel's LDS arrays.  Read with ONE scalar load.
#ifndef MXS_WIDE_TPB
#define MXS_WIDE_TPB 256
#endif
constexpr int WIDE_TPB = MXS_WIDE_TPB;  // threads of a workgroup of the wide variable launch
constexpr int WIDE_CAPB = 4 * WIDE_TPB; // staged F->V elements per block: 4 per thread
constexpr int WIDE_MAX_SLOTS = WIDE_TPB;   // outgoing edges (CSR slots) per block
constexpr int WIDE_MAX_VARS = 128;    // variables per block (their local index fits a byte)
constexpr int WIDE_MAX_COSTS = WIDE_TPB * 3 / 2;   // variables * D per block
constexpr int WIDE_MAX_D = 256;       // largest domain of the class (a row of zeros in LDS)
// (f64: 18 KB of LDS per block, eight blocks per CU: the phases of a block are chains of
// dependent loads, what hides them is the number of blocks in flight)
struct WideBlock {
    int32_t first_var;   // internal id of the first variable
    int32_t n_vars;
    int32_t D;
    int32_t slot0;       // CSR slot of its first edge (vrowptr[first_var])
    int32_t n_slots;
    uint32_t magic;      // ceil(2^32 / D): idx / D == (idx * magic) >> 32 for idx < WIDE_CAPB (D >= 2;
                         // D = 1 does not fit and is handled apart)
    int64_t cost_off;    // element offset of the first variable's own costs (vcost_off[first_var])
};

// One workgroup of the K_V_HUB class: HUB_EDGES consecutive outgoing edges (positions ko0 .. of the variable's slot range)
// of ONE variable, a lane per edge.  The lane with ko == deg (one past the last edge) is the variable's BELIEF lane: the same
// chain with nothing left out is select_value's sum (maxsum.py:607-610).  A variable of degree deg takes ceil((deg + 1) / HUB_EDGES)
// workgroups.  Everything a block needs in one record (read with ONE scalar load: no chain of dependent loads).
struct HubBlock {
    int32_t var;       // internal variable id
    int32_t ko0;       // first outgoing edge of the block
    int32_t D, deg;
    int32_t slot0;     // vrowptr[var]
    uint32_t magic;    // ceil(2^32 / ROW), ROW = 8 + min(roundup(deg, 8), HUB_TILE): e / ROW for e < 2^16 (kernels.h variable_hub)
    int64_t cost_off;  // vcost_off[var]
};
constexpr int HUB_CW = MXS_BLOCK / 128;  // pairs of waves per workgroup: one wave walks the sum_cost chains of 64 edges, its twin the
                                         // msg_costs chains of the same edges (kernels.h variable_hub)
constexpr int HUB_EDGES = 64 * HUB_CW;   // outgoing edges per workgroup
constexpr int HUB_TILE = 2048;           // edges of one value of d staged in LDS per step, at most
constexpr int HUB_LDS = 2560;            // elements of a workgroup's LDS area: rows of 8 + NK elements (kernels.h variable_hub)

struct NaryLaunch {  // one launch per (arity, nj, threads) group of K_F_NARY factors
    int32_t arity, nj;   // nj = ceil(R / BLOCK), R = product of the dimensions after the first
    int32_t threads;     // block size: ceil(R / nj) rounded up to whole waves -- when R allows
                         // it (R = 576 = 3 * 192) every lane owns exactly nj entries per d0
    int32_t first;       // first descriptor of the group
    int32_t count;
    int32_t cut;         // 1: factors reading ghost variables (second phase of a sharded cycle)
    int32_t tab_type;    // TabType the tables of the group are stored in (one kernel instantiation each)
    int32_t box;         // (>= SMALL_BASE: the small-domain kernel, small_box.h; threads = SMALL_WAVES * 64, nj = 0)
                         // 0: lane-packed / full-width kernels; else the box shape id (nary_box.h: one wave per
                         // factor, BOX_WAVES factors per workgroup; nj = 0, threads = BOX_WAVES * 64);
                         // >= BIN2_BASE: the lane grid of a binary / unary table (bin_box.h; arity 1 or 2, nj = 0)
};
// the image parameters of a factor of launch group `nl` (narrow types; lane-grid groups also at full width:
// `word` = bytes of the arithmetic type)
inline NaryPlace nary_place(const NaryLaunch& nl, const NaryDesc& d, int word = 0) {
    NaryPlace p{};
    p.box = nl.box;
    p.elem = nl.tab_type == TAB_FULL ? word : tab_elem_bytes(nl.tab_type);
    p.nt = nl.threads;
    p.slot = nary_slot_bytes(nl.nj > 0 ? nl.nj : 1, p.elem);
    p.multi = nl.box == 0 && nl.nj == NARY_NJ_MULTI;
    int64_t R = 1;
    for (int i = 1; i < (d.arity & 255); ++i) R *= d.dom[i];
    p.R = (int32_t)R;
    p.d1 = d.dom[1];
    p.d2 = d.dom[2];
    p.arity = d.arity & 255;
    for (int i = 0; i < 6; ++i) p.dom[i] = d.dom[i];
    return p;
}
// Sort code of a launch group: (box, arity, nj, waves) -- one kernel instantiation each.
constexpr int nary_group_code(int box, int arity, int nj, int waves) { return ((box * 16 + arity) * 16 + nj) * 16 + waves; }
constexpr int nary_code_box(int code) { return code >> 12; }
// A lane-grid group moves to a sibling group's wider storage type while that costs at most this many bytes per cycle (layout.cpp)
constexpr int64_t BIN2_MERGE_BYTES = (int64_t)8 << 20;
// the (nj, waves) of the lane-packed / full-width kernels for R entries per value of the first variable
constexpr int nary_classic_nj(int64_t R) { return (int)((R + BLOCK - 1) / BLOCK); }
constexpr int nary_classic_waves(int64_t R) { return (int)(((R + nary_classic_nj(R) - 1) / nary_classic_nj(R) + 63) / 64); }

struct WaveMeta {  // per wave of a K_V_PACK class: read with ONE scalar load, so a lane
                   // knows its variable and edge position without per-lane tables
    int32_t first_var;  // internal id of the variable of lane 0
    int32_t deg_nv;     // degree of the wave's variables | number of variables << 8 |
                        // ceil(2^15 / degree) << 16  (lane / degree = lane * that >> 15)
};

struct FactorGen {  // per factor of a generic / n-ary class
    int32_t edge_base;  // internal id of its first edge
    int32_t arity;
    int64_t tab_off;    // element offset of its row-major table
};

struct LayoutOptions {
    int word = 8;                // sizeof(T)
    bool no_specialise = false;  // force the generic kernels (testing)
    bool sort_by_degree = true;
    bool nary = true;            // use the workgroup-per-factor kernel (K_F_NARY)
    bool sort_factors = false;   // inside a class, factors follow their first variable's order
    bool factors_second = false; // shard: all register factor classes go to the second launch
    bool schedule = false;       // co-schedule the blocks that touch the same records (Layout::sched)
    bool compact_tables = false; // tables whose every entry a narrower type holds exactly are stored in it
    bool box = true;             // narrow arity-3 tables that fit a box shape use the one-wave-per-factor kernel
    bool pack8_fused = true;     // ... as the first workgroups of the largest lane-grid factor launch instead of a launch of their own
    bool pack8 = true;           // variables of 5..8 values and degree <= 64 use the lane-per-edge kernel (k_variable_pack8)
    bool small = true;           // arity 3..5 over domains of at most 5 values with a narrow table use the lane-group kernel (small_box.h)
    bool nary_multi = true;      // tables beyond one pass of the workgroup-per-factor kernel, and arity 6, run it in passes (generic otherwise)
    bool merge_types = true;     // small lane-grid groups of one shape share the wider sibling's storage type (and its launch)
    bool bin2 = true;            // binary / unary tables beyond the register classes use the lane-grid kernel (bin_box.h)
    bool hub = true;             // variables beyond the packed / wide classes use the wave-per-64-edges class (K_V_HUB) instead of a thread each
    bool half_cut = true;        // a shard's cut binary factors compute only the message to their own variable
    int64_t tile_bytes = -1;     // binary factors in tiled order: > 0 windows of about this many bytes, 0 never, < 0 per instance (layout.cpp)
};

struct Layout {
    LayoutOptions opt;
    int32_t n_vars = 0, n_factors = 0, n_edges = 0;
    bool is_max = false;

    // permutations (internal -> external)
    std::vector<int32_t> factor_i2e, var_i2e, edge_i2e;
    std::vector<int32_t> factor_e2i;
    // where entry k of internal factor fi lives in the device table image:
    // f_tab_base[fi] + k * f_tab_stride[fi]  (mxs_update_factor_table)
    std::vector<int64_t> f_tab_base;
    std::vector<int32_t> f_tab_stride;
    std::vector<int32_t> var_e2i, edge_e2i;

    // classes of the sweep launch (the K_F_NARY groups have their own launches)
    std::vector<ClassInfo> classes;
    std::vector<int32_t> sweep_order;  // classes of the sweep launch in launch order
    std::vector<int32_t> sweep_order2; // classes of the second sweep launch: cut factors (sharded)
    // fused sharded launch: sweep_order then sweep_order2 in ONE grid (cut classes last,
    // wait_halo set); block bases of that grid, per class of the concatenated list
    std::vector<int32_t> fused_block_base;
    int32_t n_blocks_fused = 0;        // 0: no fused launch possible (too many classes)
    int32_t n_blocks_sweep2 = 0;
    int32_t n_blocks_sweep = 0;        // grid size of launch 0
    // Block schedule of launch 0 (empty: block b works on the class whose block range holds b).
    // sched[b] = (position of the class in sweep_order) << 24 | block of that class.  The blocks
    // are ordered by where their work sits along the internal VARIABLE order -- a variable
    // block, then the factor blocks whose first scope variables it holds (sort_factors) -- and
    // every XCD gets one contiguous eighth of that order (workgroup b runs on XCD b % 8): the
    // variable side and the factor side of an edge then read the edge's two records through
    // the SAME 4-MB L2 at about the same time, and the second read of a record -- as the
    // other side's "own previous message" -- is an L2 hit instead of a second trip to the
    // Infinity Cache / HBM.  Placement is a speed matter only: any bijection gives the same
    // result.
    std::vector<uint32_t> sched;
    bool sweep_regular = false;        // the sweep has work after cycle 0 (not only isolated variables)
    std::vector<NaryDesc> ndesc;          // K_F_NARY factors, grouped by (arity, nj)
    std::vector<NaryLaunch> nary_launches;
    std::vector<int32_t> pack8_classes;   // K_V_PACK8 classes (at most one): their own launch
    std::vector<int32_t> wide_classes;    // K_V_WIDE classes (at most one: every wide variable, sorted by D)
    std::vector<WideBlock> wide_blocks;   // the workgroups of the K_V_WIDE launch
    std::vector<HubBlock> hub_blocks;     // the workgroups of the K_V_HUB class (longest chains first)

    // per internal edge (factor-major)
    std::vector<int32_t> f2v_off;    // element offset of the edge's F->V message
    std::vector<int32_t> v2f_off;    // element offset of the edge's V->F message
    std::vector<int32_t> edge_dom;   // D of the edge's variable
    std::vector<int32_t> edge_half;  // H of its messages
    std::vector<int32_t> edge_gen_factor;  // generic classes: index into fgen
    std::vector<int32_t> edge_var_int;     // internal variable id of the edge
    // 1: the send counter of the edge's F->V / V->F message lives in the first padding
    // element of the message record (register classes with H > D), not in cF / cV
    std::vector<uint8_t> edge_fcim, edge_vcim;
    int64_t f2v_elems = 0, v2f_elems = 0;  // elements per buffer
    int64_t null_f2v = 0;            // offset of an all-zero block in F2V nobody writes
                                     // (padding slots gather it: adding 0.0 is exact)

    // factors
    std::vector<FactorGen> fgen;     // generic + n-ary factors
    std::vector<double> tables;      // device image (already negated for max)
    std::vector<uint8_t> ctables;    // compact records of the classes with tab_type != TAB_FULL
                                     // (UN-negated values: the kernel flips the sign on load in max mode)
    std::vector<int64_t> f_ctab_off; // [n_factors] byte offset of the factor's compact record, -1 = none
    std::vector<int32_t> f_class;    // [n_factors] index into classes of the factor's class (-1: n-ary launch)
    std::vector<int32_t> f_ndesc;    // [n_factors] index into ndesc (K_F_NARY factors), else -1
    std::vector<uint8_t> f_tab_type; // [n_factors] TabType the factor's table is read as
    // variables (internal order)
    std::vector<int32_t> vrowptr;    // [n_vars+1] var-major slot ranges (CSR)
    std::vector<int32_t> vslot_edge; // [n_edges] internal edge id of the slot
    std::vector<int32_t> vslot_f2v;  // [n_edges] F2V offset of the slot's edge
    std::vector<int32_t> vslot_v2f;  // [n_edges] V2F offset of the slot
    std::vector<int64_t> vslot_cv;   // [n_edges] position of the slot's send counter in cV
    // per lane of the K_V_PACK classes (lane = var_in_wave * deg + k, or padding)
    std::vector<int32_t> vell;       // F2V offset of the lane's edge, -1 = padding lane
    std::vector<WaveMeta> vwave;     // one per wave (64 lanes) of the K_V_PACK classes
    int64_t n_cv = 0;                // size of the cV array (CSR slots + padded class slots)
    int dsel = 0;                    // the one D all register/wave classes share, else 0
    std::vector<int32_t> vdom;       // [n_vars]
    std::vector<int32_t> vhalf;      // [n_vars] padded message length of the variable
    std::vector<int64_t> vcost_off;  // [n_vars]
    std::vector<double> var_cost;    // device image (negated for max)
    std::vector<int32_t> init_idx;   // [n_vars] internal order, -1 = none
    std::vector<uint8_t> owned;      // [n_vars] internal order
    std::vector<uint8_t> fowned;     // [n_factors] internal order (eval_cost)

    // bookkeeping for eval_cost (row-major tables in internal factor order)
    std::vector<int64_t> eval_tab_off;  // [n_factors+1]
    std::vector<double> eval_tables;    // un-negated
    std::vector<double> eval_var_cost;  // un-negated, internal var order
    std::vector<int32_t> frowptr;       // [n_factors+1] internal

    int64_t algorithmic_bytes = 0;   // SURVEY.md section 8d formula
    bool tiled = false;              // the binary factors are in tiled order

    int half(int D) const { return half_stride(D, opt.word); }
};

// Returns "" on success, an error message otherwise.
std::string build_layout(const mxs_graph& g, const mxs_params& p, Layout& out);
// The narrowest TabType that represents every one of the n values exactly (T = `word` bytes wide).
int narrowest_tab_type(const double* v, int64_t n, int word);
// Encode one factor's `entries` values (un-negated) as a compact record of type `t` at `dst`.
void encode_tab_record(const double* v, int entries, int t, uint8_t* dst);
// One entry (un-negated) in storage type `t`; TAB_FULL = the arithmetic type of `word` bytes (lane-grid images).
void encode_tab_entry(double v, int t, int word, uint8_t* dst);
LayoutOptions options_from_params(const mxs_params& p);

}  // namespace mxs
