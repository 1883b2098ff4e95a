// layout.h -- host-side compilation of a flat factor graph (include/maxsum_gpu.h)
// into the device layout the gfx950 kernels sweep.  Pure C++, no HIP.
//
// Device layout (T = f64 or f32 element):
//
//   edge records   one record per factor-variable edge, factor-major in the
//                  engine's internal factor order, so the records of a factor
//                  are contiguous:  [ V->F message : H ][ F->V message : H ]
//                  (H = half stride >= D).  Both directions of an edge sit in
//                  one record because each side of a cycle reads both (its
//                  input and the previous message it sent, for damping and the
//                  send filter; pydcop/algorithms/maxsum.py:346-377, 537-564);
//                  the variable side's gather then touches one place per edge.
//                  Two record buffers (old/new): a cycle reads only cycle t-1
//                  (Jacobi).
//   tables         per factor class; uniform classes are stored entry-major
//                  (SoA: entry k of factor j at k*n+j) so that a wave reads
//                  them fully coalesced; generic factors keep row-major tables.
//   counters       send counters (`_prev_messages[..][1]`, maxsum.py:303,474):
//                  cF factor-major, cV variable-major -- each is private to the
//                  side that owns it, so neither is ever gathered.
//   variables      internal order = sorted by class then degree, so a wave of
//                  the register-resident variable kernel has uniform degree.
//
// Factors and variables are grouped into classes; one 256-thread block works on
// one class (BlockDesc), and one launch sweeps all classes of both sides.
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
    K_F_NARY = 4,   // large tables: workgroup per factor, LDS tile (own launch)
    K_V_REG = 5,    // D in {2,3,4}, 1 <= deg <= 4: thread per variable, registers
    K_V_GEN = 6,    // anything: thread per variable, scalar loops
    K_V_WAVE = 7,   // D in {2,3,4}, 4 < deg <= 64: G = 8, 16 or 64 lanes per variable,
                    // one lane per incoming edge, cross-lane sums
};

constexpr int BLOCK = 256;
constexpr int MAX_REG_D = 4;
constexpr int MAX_REG_DEG = 4;
constexpr int MAX_CLASSES = 24;  // ClassInfo table travels in the kernel arguments
constexpr int MAX_WAVE_DEG = 64;

struct ClassInfo {       // one per class, read with scalar loads
    int32_t kind;
    int32_t D;           // uniform domain size (0 for generic classes)
    int32_t H;           // half stride of the class's records (uniform classes)
    int32_t maxdeg;      // K_V_REG: 4; K_V_WAVE: lanes per variable G (8, 16 or 64)
    int32_t first;       // first internal factor / variable id of the class
    int32_t count;       // number of factors / variables (K_F_GEN: edges)
    int32_t edge_base;   // first internal edge id (factor classes)
    int32_t start_only;  // K_V_GEN class of degree-0 variables: only cycle 0
    int64_t rec_base;    // element offset of the first record (factor classes)
    int64_t tab_base;    // element offset of the class's tables
    int64_t cost_base;   // element offset of the class's variable costs
    int32_t block_base;  // index of the class's first block in its launch
    int32_t per_block;   // items a block covers (BLOCK, or BLOCK/G for K_V_WAVE)
    int64_t ell_base;    // K_V_REG / K_V_WAVE: first entry of the class's slot table
    int64_t cv_base;     // variable classes: first send counter of the class in cV
};

struct BlockDesc {  // n-ary launch only; the sweep derives (class, item) from blockIdx
    int32_t cls;   // index into classes
    int32_t item;  // first item (factor / edge / variable index within class)
};

struct FactorGen {  // per factor of a generic / n-ary class
    int32_t edge_base;  // internal id of its first edge
    int32_t arity;
    int64_t tab_off;    // element offset of its row-major table
};

struct LayoutOptions {
    int word = 8;             // sizeof(T)
    bool aligned_halves = true;  // each half of a record 16-byte aligned
    bool pad64 = false;          // records padded to 64 bytes when they fit
    bool no_specialise = false;  // force the generic kernels (testing)
    bool sort_by_degree = true;
    int64_t nary_min_entries = (int64_t)1 << 60;  // tables at least this big go to K_F_NARY (disabled until the LDS kernel lands)
    int64_t nary_max_bytes = 140 * 1024; // ... if they fit in LDS
};

struct Layout {
    LayoutOptions opt;
    int32_t n_vars = 0, n_factors = 0, n_edges = 0;
    bool is_max = false;

    // permutations (internal -> external)
    std::vector<int32_t> factor_i2e, var_i2e, edge_i2e;
    std::vector<int32_t> var_e2i, edge_e2i;

    // classes and blocks; launch 0 = sweep kernel, launch 1 = n-ary kernel
    std::vector<ClassInfo> classes;
    std::vector<int32_t> sweep_order;  // classes of launch 0 in launch order
    int32_t n_blocks_sweep = 0;        // grid size of launch 0
    std::vector<BlockDesc> blocks_nary;

    // per internal edge
    std::vector<int64_t> rec_off;    // element offset of the record
    std::vector<int32_t> edge_dom;   // D of the edge's variable
    std::vector<int32_t> edge_half;  // H of the record
    std::vector<int32_t> edge_gen_factor;  // generic classes: index into fgen
    std::vector<int32_t> edge_var_int;     // internal variable id of the edge
    int64_t rec_elems = 0;           // elements per record buffer

    // factors
    std::vector<FactorGen> fgen;     // generic + n-ary factors
    std::vector<double> tables;      // device image (already negated for max)
    // variables (internal order)
    std::vector<int32_t> vrowptr;    // [n_vars+1] var-major slot ranges
    std::vector<int64_t> vslot_rec;  // [n_edges] record offset of the slot's edge
    std::vector<int32_t> vslot_edge; // [n_edges] internal edge id of the slot
    std::vector<int64_t> vslot_cv;   // [n_edges] position of the slot's send counter in cV
    std::vector<int32_t> vell;       // slot tables of the K_V_REG ([4][count], edge-slot-major)
                                     // and K_V_WAVE ([count][G]) classes: record offset or -1
    std::vector<uint8_t> vdeg8;      // [n_vars] min(degree, 255), internal order
    int64_t n_cv = 0;                // size of the cV array (CSR slots + padded class slots)
    int64_t null_rec = 0;            // offset of an all-zero record nobody writes (padding
                                     // slots read it: adding 0.0 is exact)
    int dsel = 0;                    // the one D all register/wave classes share, else 0
    std::vector<int32_t> vdom;       // [n_vars]
    std::vector<int32_t> vhalf;      // [n_vars] half stride of the variable's records
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
    int max_nary_lds_bytes = 0;

    int half_stride(int D) const;
};

// Returns "" on success, an error message otherwise.
std::string build_layout(const mxs_graph& g, const mxs_params& p, Layout& out);
LayoutOptions options_from_params(const mxs_params& p);

}  // namespace mxs
