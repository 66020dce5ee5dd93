/* batch_index.h -- how stereopar-batch spreads a list of pairs over devices (SURVEY.md 8e:
 * pair j -> device j mod n_devices, no exchange between devices).  A header of its own so that
 * the mapping is testable without a GPU (tests/test_batch_index_cpu.py). */
#ifndef BATCH_INDEX_H
#define BATCH_INDEX_H

/* number of pairs of a list of n_pairs that device `rank` of n_devices owns */
static inline int batch_pairs_of_rank(int n_pairs, int rank, int n_devices)
{
    return rank < n_pairs ? (n_pairs - rank + n_devices - 1) / n_devices : 0;
}

/* global index (line of the pair list) of the seq-th pair device `rank` processes; with a
 * repeat count the device walks its `mine` pairs round and round (seq runs to mine * repeat) */
static inline int batch_global_index(int rank, long seq, int mine, int n_devices)
{
    return rank + (int)(seq % mine) * n_devices;
}

#endif
