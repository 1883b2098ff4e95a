/*
 * maxsum_gpu.h -- C-ABI of the MI355X-native synchronous Max-Sum engine.
 *
 * This is the drop-in boundary for the hot path of
 * pydcop/algorithms/maxsum.py (reference @ /root/reference).  The reference is
 * pure Python and has no FFI of its own; the entry points below are what a
 * ctypes binding for `pydcop.algorithms.maxsum_gpu` binds (INTEGRATION.md shows
 * the stub).  Each entry point cites the reference code it replaces.
 *
 * Conventions
 *   - every call returns 0 on success, a negative MXS_E_* code otherwise and
 *     never throws across the ABI; mxs_last_error() gives the message of the
 *     last failure on the calling thread;
 *   - host buffers are caller-owned (numpy arrays); device buffers are owned
 *     by the engine; a handle is NOT thread-safe (one owning thread, like a
 *     pyDCOP computation: pydcop/infrastructure/computations.py:277-282);
 *   - there is no CPU fallback: without a usable gfx950 device mxs_create
 *     fails with MXS_E_NODEVICE.
 *
 * Flat factor-graph format (the "compiled" DCOP; indices, never domain values)
 *   variables  v = 0..n_vars-1      dom_size[v] = |D_v|
 *   factors    f = 0..n_factors-1   scope = edge_var[factor_rowptr[f] ..
 *                                   factor_rowptr[f+1]) in `factor.dimensions`
 *                                   order (pydcop/dcop/relations.py:682-690)
 *   edges      e = 0..n_edges-1     factor-major: edge e is position
 *                                   e-factor_rowptr[f] of factor f
 *   tables     tables[table_off[f] ...] row-major over the scope, C order
 *                                   (NAryMatrixRelation._m, relations.py:716-733)
 *   var side   var_edges[var_rowptr[v] .. var_rowptr[v+1]) = edge ids in
 *                                   VariableComputationNode.links order
 *                                   (pydcop/computations_graph/factor_graph.py:126-128)
 *   messages   message of edge e occupies msg_off(e) .. msg_off(e)+D_{var(e)}
 *              with msg_off = exclusive prefix sum of dom_size[edge_var[e]]
 */
#ifndef MAXSUM_GPU_H
#define MAXSUM_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MXS_OK            0
#define MXS_E_INVALID    -1  /* bad argument / malformed graph               */
#define MXS_E_NODEVICE   -2  /* no gfx950 device / HIP runtime unusable      */
#define MXS_E_HIP        -3  /* a HIP call failed (see mxs_last_error)       */
#define MXS_E_NOMEM      -4
#define MXS_E_STATE      -5  /* call not valid in the engine's current state */
#define MXS_E_COMM       -6  /* RCCL missing or a RCCL call failed               */

/* dcop.objective, AlgorithmDef.mode (pydcop/algorithms/__init__.py:141) */
#define MXS_MODE_MIN 0
#define MXS_MODE_MAX 1
/* algo_params "damping_nodes" (pydcop/algorithms/maxsum.py:214-216) */
#define MXS_DAMP_NONE    0
#define MXS_DAMP_VARS    1
#define MXS_DAMP_FACTORS 2
#define MXS_DAMP_BOTH    3
/* algo_params "start_messages" (pydcop/algorithms/maxsum.py:219) */
#define MXS_START_LEAFS      0
#define MXS_START_LEAFS_VARS 1
#define MXS_START_ALL        2
/* arithmetic type of messages/tables on the device */
#define MXS_DTYPE_F64 0   /* the reference's arithmetic (python float)       */
#define MXS_DTYPE_F32 1   /* throughput mode                                 */

typedef struct mxs_graph {
    int32_t        n_vars;
    int32_t        n_factors;
    int32_t        n_edges;
    const int32_t *dom_size;       /* [n_vars]                                */
    const double  *var_cost;       /* [sum dom_size] Variable.cost_for_val,
                                      pydcop/dcop/objects.py:231,429,498      */
    const int32_t *init_idx;       /* [n_vars] index of Variable.initial_value
                                      or -1; may be NULL (= all -1)           */
    const int32_t *factor_rowptr;  /* [n_factors+1]                           */
    const int32_t *edge_var;       /* [n_edges]                               */
    const int64_t *table_off;      /* [n_factors+1]                           */
    const double  *tables;         /* [table_off[n_factors]]                  */
    const int32_t *var_rowptr;     /* [n_vars+1]                              */
    const int32_t *var_edges;      /* [n_edges]                               */
    const uint8_t *var_owned;      /* [n_vars] 1 = this engine updates the
                                      variable, 0 = ghost copy of a variable
                                      owned by another shard (its V->F
                                      messages arrive through mxs_halo_*).
                                      NULL = all owned (single GPU).          */
    const uint8_t *factor_owned;   /* [n_factors] 1 = mxs_eval_cost counts the
                                      factor on this shard (a cut factor is
                                      replicated on every shard that owns one
                                      of its variables but must be counted
                                      once).  NULL = all.                     */
    const double  *eval_var_cost;  /* [sum dom_size] variable costs WITHOUT the
                                      Max-Sum noise, what mxs_eval_cost sums:
                                      the reference adds its noise inside the
                                      computation only (VariableNoisyCostFunc,
                                      maxsum.py:476-487) while DCOP.solution_cost
                                      (dcop.py:308-367) reads the variables' own
                                      cost functions.  NULL = var_cost (no noise
                                      folded in).                              */
} mxs_graph;

typedef struct mxs_params {
    int32_t mode;            /* MXS_MODE_*                                    */
    int32_t damping_nodes;   /* MXS_DAMP_*                                    */
    int32_t start_messages;  /* MXS_START_*                                   */
    int32_t dtype;           /* MXS_DTYPE_*                                   */
    double  damping;         /* maxsum.py:213, default 0.5                    */
    double  stability;       /* maxsum.py:217, default 0.1                    */
    int32_t graph_chunk;     /* cycles captured per hipGraph replay; 0 = eager
                                launches; <0 = engine default: replay tiny
                                graphs (< 4 MB per cycle), launch the rest
                                eagerly (replay slows long kernels down)      */
    int32_t layout_flags;    /* 0 = default.  Testing / measurement knobs:
                                bit2 (4)  generic kernels only (no specialised classes)
                                bit3 (8)  keep the caller's variable order where possible
                                bit4 (16) no workgroup-per-factor kernel
                                bit5 (32) variable side only, bit6 (64) factor side
                                          only: TIMING ONLY, results are wrong
                                bit7 / bit8 (128 / 256) factors of a class sorted by their
                                          first variable / in the caller's order
                                bit9 / bit10 (512 / 1024) shard: every factor class in the
                                          second launch / only the cut classes
                                bit11 / bit12 (2048 / 4096) block schedule off / on
                                bit13 / bit14 (8192 / 16384) full-width / compact tables
                                bit15 (32768) no one-wave-per-factor box kernel (arity-3
                                          integer tables keep the lane-packed kernel)
                                bit16 (65536) shard: cut binary factors compute both
                                          messages (default: only the one to their own variable)
                                bit17 / bit18 (131072 / 262144) tiled order of the binary
                                          factors always / never (default: per instance --
                                          4-byte words or a cache-resident cycle, and a
                                          variable order that is not local already).  Where
                                          neither bit is set, $MAXSUM_TILE_KB = <kilobytes>
                                          overrides the window size (0 = off; A/B runs and the
                                          parity tests of the tiled order only)
                                bit19 (524288) no lane-grid kernel for binary / unary factors
                                          beyond the register classes (thread per edge instead)
                                bit20 (1048576) no lane-per-edge kernel for variables of 5..8 values
                                          (the workgroup-per-run kernel of the wide class instead)
                                bit21 (2097152) that kernel as a launch of its own (default: its workgroups
                                          are the first ones of the largest lane-grid factor launch)
                                bit23 (8388608) no small-domain lane-group kernel for factors of arity 3..5
                                          (the workgroup-per-factor kernels instead; A/B runs)
                                bit22 (4194304) no hub class: variables beyond the packed / wide classes
                                          take one thread each (the round-5 behaviour; A/B runs)
                                bit24 (16777216) lane-grid groups of one shape stay split by storage type
                                          (default: a small group takes the wider sibling's type and
                                          rides in its launch; A/B runs)
                                bit25 (33554432) no multi-pass workgroup-per-factor kernel: tables of more than
                                          1 024 entries per value of the first variable (arity 3 over more
                                          than 32 values, ...) and arity 6 take a thread per edge (A/B runs)   */
} mxs_params;

typedef struct mxs_engine mxs_engine;

/* Number of visible HIP devices (0 and MXS_OK when the runtime loads but
 * sees no GPU). */
int mxs_device_count(int32_t *count);

/* Build an engine on HIP device `device`, upload the graph and run cycle 0
 * (`start()`): replaces building one MaxSum{Factor,Variable}Computation per
 * node (maxsum.py:118-124, 279-303, 450-487) and their on_start
 * (maxsum.py:305-328, 495-523; computations.py:741-753). */
int mxs_create(const mxs_graph *g, const mxs_params *p, int32_t device,
               mxs_engine **out);

/* Back to the state right after cycle 0. */
int mxs_reset(mxs_engine *e);

/* Run `n_cycles` synchronous cycles: every factor's and every variable's
 * on_new_cycle exactly n_cycles times (maxsum.py:339-379, 525-565; the BSP
 * barrier of computations.py:684-788 becomes the kernel boundary).  Returns
 * after the device has finished. */
int mxs_run(mxs_engine *e, int32_t n_cycles);

/* Same, and report the device time of the n_cycles sweeps measured with HIP
 * events recorded on the engine's own stream (milliseconds). */
int mxs_run_timed(mxs_engine *e, int32_t n_cycles, float *elapsed_ms);

/* `reps` repetitions of `n_cycles` cycles enqueued back to back, one HIP event between two repetitions
 * and ONE host wait at the end; elapsed_ms[r] = device time of repetition r.  For benchmarks: a
 * median over a timed region long enough for the clocks to settle, when one repetition is a fraction
 * of a millisecond (bench.py).  Not on a sharded engine. */
int mxs_run_reps(mxs_engine *e, int32_t n_cycles, int32_t reps, float *elapsed_ms);

/* Enqueue `n_cycles` cycles on the engine's stream without waiting. */
int mxs_run_async(mxs_engine *e, int32_t n_cycles);
int mxs_sync(mxs_engine *e);

/* Number of on_new_cycle calls done so far (computations.py:790-792). */
int mxs_cycle_count(const mxs_engine *e, int64_t *cycles);

/* Current selection: idx[v] = index in the domain of the selected value and
 * belief[v] = its cost, i.e. (current_value, current_cost) set through
 * value_selection(*select_value(...)) (maxsum.py:532, 584-620;
 * computations.py:1058-1078).  Either pointer may be NULL. */
int mxs_get_assignment(mxs_engine *e, int32_t *idx, double *belief);

/* Parity/debug: the messages every receiver currently holds, in the caller's
 * edge order (msg_off layout above), and the per-directed-edge send counters
 * (`_prev_messages[...][1]`, maxsum.py:303,474).  Any pointer may be NULL.
 * On a SHARD (var_owned given) the state of the edges towards ghost variables is not this shard's
 * to hold: a cut binary factor computes only the message to its own variable (layout_flags bit16
 * off, the default), so the F->V record and send counter of its edge to the ghost variable stay
 * zero here -- the shard that owns that variable holds them -- and a ghost variable's V->F record
 * is whatever the last exchange delivered.  The same applies to what mxs_set_state restores. */
int mxs_get_messages(mxs_engine *e, double *v2f, double *f2v,
                     uint8_t *count_v2f, uint8_t *count_f2v);

/* The inverse of mxs_get_messages + mxs_get_assignment: put the engine into the state
 * described by the caller's arrays (same layouts), as if `cycles` cycles had produced it.
 * Checkpoint / resume of a run, and the way a run survives a change of the graph itself
 * (maxsum_dynamic.py:234-271, DynamicFactorComputation.change_factor_function with a new
 * scope; :352-405 REMOVE / ADD on the variable side): the host builds the new flat graph, a
 * new engine, and carries the messages of the surviving edges over (pydcop_amd/dynamic.py).
 * NULL = leave that part as it is.  Not available on a shard with an exchange set up. */
int mxs_set_state(mxs_engine *e, const double *v2f, const double *f2v,
                  const uint8_t *count_v2f, const uint8_t *count_f2v,
                  const int32_t *idx, const double *belief, int64_t cycles);

/* DCOP.solution_cost (pydcop/dcop/dcop.py:308-367): sum of factor costs and
 * variable costs of an assignment; a term exactly equal to `infinity` counts
 * as one violation instead.  idx == NULL evaluates the current selection. */
int mxs_eval_cost(mxs_engine *e, const int32_t *idx, double infinity,
                  double *cost, int64_t *violations);

/* Algorithmic bytes moved by one cycle (SURVEY.md section 8d formula) and the
 * number of kernel launches per cycle -- used by bench.py for the roofline. */
int mxs_cycle_bytes(const mxs_engine *e, int64_t *algorithmic_bytes,
                    int32_t *launches_per_cycle);

/* How the cost tables are stored on the device: factors[t] = number of factors whose table is
 * kept as t = 0 full width (the engine's arithmetic type), 1 f32, 2 int16, 3 int8, and the table
 * bytes one cycle reads.  A register class (unary / binary factors, D <= 4) whose every entry
 * is exactly representable in a narrower type is stored in it and widened on load: lossless --
 * the arithmetic and every result are bit for bit those of full-width tables -- and up to 4.5x
 * fewer table bytes per cycle.  An update that does not fit (mxs_update_factor_table,
 * mxs_set_parent_table) moves the class back to full width.  Likewise per factor for the
 * workgroup- / wave-per-factor kernels (lane-packed slots, box records) and for the lane-grid
 * kernel of binary / unary tables, whose image -- the row-major table cut into lane pieces -- is
 * counted at every width. */
int mxs_table_storage(const mxs_engine *e, int64_t factors[4], int64_t *table_bytes_per_cycle);

/* Order of the binary factors inside their classes: *tiled = 1 when they are grouped by (window of the
 * first variable, window of the second) so that every gather of a tile falls in two L2-sized windows,
 * 0 when they simply follow their first variable.  A layout decision only (layout_flags bit17 / bit18,
 * default per instance): any order computes the same messages bit for bit. */
int mxs_factor_order(const mxs_engine *e, int32_t *tiled);

/* Which kernel computes factor_costs_for_var (maxsum.py:382-447) for how many factors:
 * counts[0] register class, unary (thread per factor, D <= 4); [1] register class, binary (D x D, D <= 4);
 * [2] generic (thread per edge, scalar loops: whatever nothing else takes); [3] workgroup per factor
 * (arity 2..5, 64..1024 entries per value of the first variable; full-width or lane-packed tables);
 * [4] one wave per factor (arity 3, integer tables in box records); [5] lane grid per factor (binary /
 * unary tables beyond the register classes, up to 64 x 64: 4 / 16 / 64 lanes per factor); [6] lane group per factor
 * (round 6: arity 3..5, every domain at most 5 values, a narrow table: 8 / 32 lanes per factor, small_box.h).
 * A layout decision only: every kernel computes the same messages bit for bit. */
int mxs_factor_kernels(const mxs_engine *e, int64_t counts[7]);

/* Which kernel runs on_new_cycle of how many variables (maxsum.py:525-565): counts[0] packed class (lane per
 * edge, D <= 4, degree <= 64: part of the sweep launch); [1] the same scheme on 8-element records (5 <= D <= 8;
 * own launch); [2] wide class (a workgroup per run of variables of one domain size, messages staged in LDS);
 * [3] generic (thread per variable); [4] not swept (isolated variables after cycle 0, a shard's ghosts);
 * [5] hub class (round 6: degree above 64 on a domain of at most 8 values, above 256 on any, deg * D > 1024 -- a
 * wave per 64 outgoing edges, a lane per edge; part of the sweep launch). */
int mxs_variable_kernels(const mxs_engine *e, int64_t counts[6]);

/* Replace the cost table of factor `factor` (caller's factor index) by one of the
 * same shape, row-major over its scope; messages, counters and the selection
 * carry on from where they are: change_factor_function of
 * pydcop/algorithms/maxsum_dynamic.py:80-104 without rebuilding the graph. */
int mxs_update_factor_table(mxs_engine *e, int32_t factor, const double *table, int64_t n_entries);

/* Factors whose relation also depends on external (read-only) variables
 * (maxsum_dynamic.py:113-186 FactorWithReadOnlyVariableComputation, :188-232 and :273-288
 * DynamicFactorComputation): the engine keeps the whole relation -- `parent`, row-major over
 * `n_dims` dimensions of sizes `dims`, `is_external[i]` = 1 for a read-only dimension; the other
 * dimensions, in order, are the factor's scope -- on the device, and
 * mxs_slice_factor(external_idx: one value index per external dimension, in order) makes the
 * factor's active table the slice at those values (relation.slice, relations.py:760-810) with
 * one small kernel: no table crosses PCIe when a sensor value changes.  Like
 * mxs_update_factor_table the messages carry on (the reference swaps `self._factor` and goes
 * on, maxsum_dynamic.py:100-104). */
int mxs_set_parent_table(mxs_engine *e, int32_t factor, const double *parent, int32_t n_dims,
                         const int32_t *dims, const uint8_t *is_external);
int mxs_slice_factor(mxs_engine *e, int32_t factor, const int32_t *external_idx);

/* Profiling only: run ONE more cycle in which every block of the sweep launch
 * records {start, end} (wall_clock64 ticks, 100 MHz) and its class kind;
 * out[3*b .. 3*b+2] for block b, `cap` = blocks the buffer can hold.  out == NULL
 * only reports the number of blocks. */
int mxs_debug_timeline(mxs_engine *e, int64_t *out, int32_t cap, int32_t *n_blocks);

/* ---- sharded (multi-GPU) operation: one engine per rank ----------------
 * A shard's graph holds its owned variables, every factor touching one of
 * them, and ghost copies (var_owned = 0) of the remote variables those cut
 * factors touch.  After each cycle the owner's new V->F messages of cut edges
 * are copied into the ghost slots of the other shard.  The engine only packs
 * and unpacks; the exchange itself (RCCL all-to-all) is done by the host on
 * the device buffers returned here. */
int mxs_halo_setup(mxs_engine *e, const int32_t *send_edges, int64_t n_send,
                   const int32_t *recv_edges, int64_t n_recv);
/* Device pointers + sizes in bytes of the packed send / receive staging
 * buffers (element type = the engine dtype; a message of edge e takes
 * dom_size[edge_var[e]] elements, edges in the order given to halo_setup). */
int mxs_halo_buffers(mxs_engine *e, void **send_dev, int64_t *send_bytes,
                     void **recv_dev, int64_t *recv_bytes);
/* Use caller-owned device memory (e.g. tensors of the framework that runs the
 * collective) as the packed send / receive buffers instead of the engine's own;
 * sizes as reported by mxs_halo_buffers.  Call after mxs_halo_setup; the
 * current messages are packed into the new send buffer. */
int mxs_halo_bind(mxs_engine *e, void *send_dev, void *recv_dev);
/* One sharded cycle, split so that the exchange of cycle t hides behind the part
 * of cycle t+1 that does not need it.  Per cycle the host calls, in this order:
 *   mxs_step_compute  compute stream: variables + interior factors of cycle t,
 *                     then (after the unpack of cycle t-1) the cut factors --
 *                     the only readers of ghost messages; comm stream: waits
 *                     for the variables of cycle t, gathers the owned cut-edge
 *                     V->F messages into the send buffer (mxs_step_pack does
 *                     only that part, e.g. after mxs_halo_bind);
 *   <collective>      enqueued by the host on the comm stream (mxs_stream);
 *   mxs_step_unpack   comm stream: scatters the received messages into the
 *                     ghost slots.
 * Nothing blocks the host; mxs_sync waits for both streams. */
int mxs_step_compute(mxs_engine *e);
int mxs_step_pack(mxs_engine *e);
int mxs_step_unpack(mxs_engine *e);
/* The engine's comm hipStream_t (the stream the collective has to be enqueued
 * on), as an opaque pointer. */
int mxs_stream(mxs_engine *e, void **stream);

/* ---- native exchange: the engine calls RCCL itself ------------------------
 * Replaces the agents' message transport for boundary messages
 * (pydcop/infrastructure/communication.py:588-698) without an interpreter
 * between two cycles: one exchange = one group of ncclSend / ncclRecv per
 * peer (an all-to-all with the fixed counts of the partition) on the comm
 * stream.  RCCL is dlopen'ed from `rccl_path` (NULL = "librccl.so"); a process
 * must hold ONE copy, so pass the one torch bundles when torch is loaded. */
#define MXS_UNIQUE_ID_BYTES 128
/* ncclGetUniqueId: called by rank 0, the launcher hands the bytes to every
 * other rank (torch.distributed store, MPI, a file ...). */
int mxs_comm_unique_id(const char *rccl_path, uint8_t *id /* [128] */);
/* ncclCommInitRank on the engine's device (collective over all ranks).  Call
 * after mxs_halo_setup.  send_counts[q] / recv_counts[q] = ELEMENTS of the packed
 * send / receive buffer exchanged with rank q, peers in ascending order (the
 * order of the edge lists given to mxs_halo_setup); [rank] must be 0 when world > 1. */
int mxs_comm_init(mxs_engine *e, const char *rccl_path, int32_t rank, int32_t world,
                  const uint8_t *id, const int64_t *send_counts, const int64_t *recv_counts);
/* Enqueue one exchange on the comm stream (between pack and unpack; used for the
 * start messages of cycle 0 and after mxs_reset). */
int mxs_comm_exchange(mxs_engine *e);
/* n sharded cycles, each mxs_step_compute -> exchange -> mxs_step_unpack, enqueued
 * by the library; does not wait (mxs_sync does). */
int mxs_run_sharded(mxs_engine *e, int32_t n_cycles);
/* ---- peer-store exchange: no collective ------------------------------------
 * The ranks of one node (2..8) map each other's ghost buffers and flag words through
 * hipIpc; from then on the variable kernel stores the records of cut edges straight
 * into the ghost region of the shard that holds the factor's replica (xGMI peer
 * stores), a tiny kernel behind every launch publishes the launch number in the
 * peers' flag words, and a cycle is ONE launch whose cut factor blocks poll those
 * words (time-limited; a wait that expires is reported by mxs_sync).  What the
 * agents' transport (pydcop/infrastructure/communication.py:588-698) does with one
 * message per edge and cycle is then a 32-byte store.
 *   mxs_peer_export   after mxs_halo_setup, instead of mxs_comm_init: allocates the
 *                     ghost regions and describes them.  out->qualifies == 0: this
 *                     shard cannot run in this mode (cut factors that are not binary
 *                     register classes, sent edges outside the packed variable
 *                     classes or bound for two shards, too many cut blocks, or
 *                     MAXSUM_SHARD_P2P=0) -- use mxs_comm_init then.
 *   mxs_peer_connect  once every rank holds every rank's mxs_peer_info -- the launcher
 *                     gathers them -- and ALL qualify: maps the peers' buffers, pushes
 *                     the current messages.  mxs_run_sharded / mxs_step_compute then
 *                     run fused launches; mxs_comm_exchange / mxs_step_unpack are no-ops.
 *                     mxs_reset needs a barrier over the ranks before it (no peer may
 *                     still be running cycles of the previous run). */
#define MXS_MAX_PEERS 8
/* Largest arity of a factor every engine of the library accepts (a table of more than 2^31 entries
 * cannot be addressed: 30 binary variables; the reference has no limit, maxsum.py:411-421). */
#define MXS_MAX_ARITY 30
#define MXS_IPC_HANDLE_BYTES 64
typedef struct mxs_peer_info {
    int32_t qualifies;
    int32_t rank;
    int64_t ghost_len;                    /* elements of one ghost region          */
    int64_t recv_at[MXS_MAX_PEERS];       /* where rank q's block starts in it ... */
    int64_t recv_len[MXS_MAX_PEERS];      /* ... and its length (elements)         */
    uint8_t ghost_handle[MXS_IPC_HANDLE_BYTES];
    uint8_t flag_handle[MXS_IPC_HANDLE_BYTES];
    int64_t pid;                          /* ranks of ONE process (k shards on one GPU,  */
    uint64_t ghost_ptr, flag_ptr;         /* tests, tools) use the pointers themselves   */
} mxs_peer_info;
int mxs_peer_export(mxs_engine *e, int32_t rank, int32_t world, const int64_t *send_counts,
                    const int64_t *recv_counts, mxs_peer_info *out);
int mxs_peer_connect(mxs_engine *e, const mxs_peer_info *all /* [world] */);

/* How this shard runs its cycles (decided by mxs_halo_setup / mxs_comm_init from the
 * shape of the shard):
 *   direct_exchange  2: peer stores (above).  1: the variable kernel writes the records of cut edges into the send
 *                    buffer itself and RCCL receives straight into the ghost slots (laid out
 *                    in receive order) -- no pack / unpack kernel; needs mxs_comm_init, every
 *                    sent edge in a packed variable class and sent to one shard only.
 *                    MAXSUM_SHARD_DIRECT=0 keeps the pack / unpack kernels.
 *                    "Packed variable class" = the lane-per-edge class of domains of at most 4 values: a
 *                    shard whose BOUNDARY variables have 5..8 values (the lane-per-edge class on 8-element
 *                    records), wider domains or hub degrees reports 0 here and exchanges through the pack /
 *                    unpack kernels -- correct, one launch more on each side of the collective.
 *   fused_launch     1: one sweep launch per cycle whose last blocks (the cut factors) wait
 *                    for the halo inside the kernel (opt-in: MAXSUM_SHARD_FUSED=1; measured
 *                    slower than the two-launch schedule). */
int mxs_shard_mode(const mxs_engine *e, int32_t *fused_launch, int32_t *direct_exchange);

int mxs_destroy(mxs_engine *e);

/* Message of the last error on this thread ("" if none). */
const char *mxs_last_error(void);

/* ---- asynchronous Max-Sum (pydcop/algorithms/amaxsum.py) -------------------------------------
 * The reference's amaxsum runs one handler per DELIVERED message (factor: amaxsum.py:191-250,
 * waits until every variable has been heard from, answers everyone but the sender; variable:
 * :366-424, selects its value and answers every factor but the sender), so what it computes
 * depends on the delivery order.  This engine reproduces it under the one order that is defined
 * without a thread scheduler: every computation started in graph order (variables, then
 * factors) and ONE first-in-first-out queue -- handled a GENERATION at a time (generation 0 = the
 * start messages, generation g + 1 = the messages sent while handling generation g; a FIFO
 * handles all of g before any of g + 1).  Bit for bit the reference's own computations under that
 * order (oracle/amaxsum_oracle.c, pinned by tests/test_amaxsum_oracle_vs_reference.py).  The run
 * ends by itself when the send rule (approx_match + SAME_COUNT, amaxsum.py:222-244) has silenced
 * every edge.  Same flat graph and parameters as mxs_create; messages in the msg_off layout. */
typedef struct mxs_amaxsum mxs_amaxsum;
/* start() of every computation (amaxsum.py:140-160, 295-332): generation 0 is queued. */
int mxs_amaxsum_create(const mxs_graph *g, const mxs_params *p, int32_t device, mxs_amaxsum **out);
int mxs_amaxsum_reset(mxs_amaxsum *e);
/* Deliver whole generations while the next one's number is < max_generations (< 0: until no
 * message is left); *delivered = messages handled by this call. */
int mxs_amaxsum_run(mxs_amaxsum *e, int32_t max_generations, int64_t *delivered);
/* Number of the generation waiting in the queue, its size (0 = quiescent), messages handled so far. */
int mxs_amaxsum_status(const mxs_amaxsum *e, int32_t *next_generation, int64_t *pending, int64_t *delivered);
/* Messages per generation so far (the waiting one included); *n = number of generations. */
int mxs_amaxsum_generation_sizes(const mxs_amaxsum *e, int64_t *out, int32_t cap, int32_t *n);
/* (current_value index, current_cost) of every variable (value_selection, amaxsum.py:381-383). */
int mxs_amaxsum_get_assignment(mxs_amaxsum *e, int32_t *idx, double *belief);
/* Parity/debug: per edge, what the factor holds from the variable (f_cost, f_has) and last sent
 * to it (f_prev, f_cnt = _prev_messages count), and the same on the variable side. */
int mxs_amaxsum_get_messages(mxs_amaxsum *e, double *f_cost, double *v_cost, double *f_prev, double *v_prev,
                             uint8_t *f_has, uint8_t *v_has, uint8_t *f_cnt, uint8_t *v_cnt);
int mxs_amaxsum_eval_cost(mxs_amaxsum *e, const int32_t *idx, double infinity, double *cost, int64_t *violations);
/* DynamicFunctionFactorComputation.change_factor_function with the same scope
 * (pydcop/algorithms/maxsum_dynamic.py:80-104 -- the reference defines it over the ASYNCHRONOUS
 * factor computation): `self.factor = fn`, nothing is sent; the new table (row-major in the
 * factor's own dimension order, n_entries = its size) is what the deliveries from now on are
 * computed with, held costs and last-sent messages carry on. */
int mxs_amaxsum_update_factor_table(mxs_amaxsum *e, int32_t factor, const double *table, int64_t n_entries);
int mxs_amaxsum_destroy(mxs_amaxsum *e);

/* ---- MGM (pydcop/algorithms/mgm.py) on the same flat arrays ------------------------------------
 * Factors = the constraints of the constraints hypergraph (mgm.py:68), variables = the MGM
 * computations.  One round = every variable's `_handle_value_message` once all values are in
 * (mgm.py:335-391: cost of the current value on the first round, best unilateral move, gain) and
 * `_handle_gain_message` once all gains are in (:499-540: the largest gain of a neighbourhood moves,
 * ties by name :566-588) -- n rounds = the reference with stop_cycle = n + 1.  `name_rank[v]` = rank
 * of the variable's name in sorted order (NULL: index order).  The reference's draws from the
 * unseeded `random` module are fixed: first domain value at start (unless init_idx), first of
 * equally good values.  Bit for bit the reference's own MgmComputation objects under those
 * choices (oracle/mgm_oracle.c, pinned by tests/test_mgm_oracle_vs_reference.py). */
typedef struct mxs_mgm mxs_mgm;
int mxs_mgm_create(const mxs_graph *g, const mxs_params *p, const int32_t *name_rank, int32_t device,
                   mxs_mgm **out);
int mxs_mgm_reset(mxs_mgm *e);
/* The order of every variable's domain VALUES: value_rank[cost_off(v) + d] = position of the d-th value
   of v's domain among v's values in ascending order (cost_off = prefix sum of dom_size).  Used for ONE
   thing: a variable without neighbours starts at the optimum of its own costs, and the reference breaks
   cost ties there on the VALUE -- optimal_cost_value takes min / max over (cost, value) tuples
   (pydcop/dcop/relations.py:1661-1665).  NULL or never called: the values are in ascending order as
   written (rank = index).  Resets the engine. */
int mxs_mgm_set_value_rank(mxs_mgm *e, const int32_t *value_rank);
int mxs_mgm_run(mxs_mgm *e, int32_t n_rounds);
int mxs_mgm_rounds(const mxs_mgm *e, int64_t *rounds);
/* current value index, the cost the computation holds (has_cost = 0: still None, mgm.py:349),
 * last gain and the move it would make; any pointer may be NULL */
int mxs_mgm_get_state(mxs_mgm *e, int32_t *idx, double *cost, uint8_t *has_cost, double *gain, int32_t *new_value);
int mxs_mgm_eval_cost(mxs_mgm *e, const int32_t *idx, double infinity, double *cost, int64_t *violations);
int mxs_mgm_destroy(mxs_mgm *e);

/* ---- DSA (pydcop/algorithms/dsa.py, variants A / B / C) on the same flat arrays --------------
 * One cycle = every variable's `evaluate_cycle` once all its neighbours' values are in
 * (dsa.py:319-359: best values and their cost, delta, the variant's rule :361-409, the
 * probabilistic move :411-419) -- n cycles = the reference with stop_cycle = n.  variant: 0 = A,
 * 1 = B, 2 = C; `probability` and p_mode arity (1.2 / sum(arity - 1), dsa.py:256-259) as the
 * reference's parameters.  The reference draws from Python's unseeded `random`; here every draw
 * comes from a counter-based generator keyed on (seed, variable, cycle, draw), the same function
 * on the device, in the oracle and -- patched into the reference for the duration of a run -- in
 * the pinning harness: bit for bit the reference's own DsaComputation objects under that generator
 * (oracle/dsa_oracle.c, tests/test_dsa_oracle_vs_reference.py), independent of scheduling. */
typedef struct mxs_dsa mxs_dsa;
int mxs_dsa_create(const mxs_graph *g, const mxs_params *p, int32_t variant, double probability,
                   int32_t arity_mode, uint64_t seed, int32_t device, mxs_dsa **out);
int mxs_dsa_reset(mxs_dsa *e);
/* as mxs_mgm_set_value_rank (dsa.py:278-289 calls the same optimal_cost_value) */
int mxs_dsa_set_value_rank(mxs_dsa *e, const int32_t *value_rank);
int mxs_dsa_run(mxs_dsa *e, int32_t n_cycles);
int mxs_dsa_cycles(const mxs_dsa *e, int64_t *cycles);
/* current value index and the cost the computation holds (0 until its first move) */
int mxs_dsa_get_state(mxs_dsa *e, int32_t *idx, double *cost);
int mxs_dsa_eval_cost(mxs_dsa *e, const int32_t *idx, double infinity, double *cost, int64_t *violations);
int mxs_dsa_destroy(mxs_dsa *e);

/* Library/ABI version (major*100+minor). */
int32_t mxs_version(void);

/* What this binary is: 1 = the product, compiled by hipcc for gfx950; 0 = the host
 * emulation of the very same sources that the CPU tests build (tests/emu, g++ against a
 * fake HIP runtime).  pydcop_amd/engine.py refuses a library of kind 0 unless a test has
 * registered it from Python: the product has no CPU path, by environment variable or
 * otherwise. */
int32_t mxs_build_kind(void);

#ifdef __cplusplus
}
#endif
#endif /* MAXSUM_GPU_H */
