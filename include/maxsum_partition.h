/*
 * maxsum_partition.h -- C-ABI of the host-side graph partitioner of the sharded
 * (multi-GPU) Max-Sum sweep (libmxs_partition.so; plain C++, no GPU).
 *
 * Reference counterpart: the distribution of computations on agents,
 * the modules of pydcop/distribution (e.g. the ILP of distribution/ilp_fgdp.py:85-277 minimises
 * communication between agents under capacity constraints); here the objective is
 * balanced parts with few cut factors, because every cut factor is replicated and
 * its remote variables' V->F messages cross once per cycle.
 */
#ifndef MAXSUM_PARTITION_H
#define MAXSUM_PARTITION_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Multilevel k-way partition (heavy-edge matching, greedy growing, boundary FM,
 * recursive bisection) of the variables of a factor graph in the flat format of
 * maxsum_gpu.h (factor_rowptr / edge_var).  part[v] in 0..k-1.  Vertex weight =
 * 1 + degree; `imbalance` (>= 1, e.g. 1.03) bounds a part's weight relative to the
 * mean.  Deterministic for a given seed.  Returns 0, or a negative code
 * (mxp_last_error gives the message). */
int mxp_partition(int32_t n_vars, int32_t n_factors, const int32_t *factor_rowptr,
                  const int32_t *edge_var, int32_t k, double imbalance, uint64_t seed,
                  int32_t *part /* [n_vars] */);

const char *mxp_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MAXSUM_PARTITION_H */
