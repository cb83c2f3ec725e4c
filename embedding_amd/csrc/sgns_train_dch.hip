// sgns_train_dch.hip — the trainer kernels of sgns_kernels.h for ONE row width: compiled with -DDGE_DCH=n (n = 64-float chunks
// per row: 1, 2, 3, 4, 6, 8) into sgns_dch<n>.o; sgns.hip calls dge_launch_train_dch<n>.
#include "sgns_kernels.h"

#ifndef DGE_DCH
#error "compile with -DDGE_DCH=<chunks per row>"
#endif
#define DGE_CAT_(a, b) a##b
#define DGE_CAT(a, b) DGE_CAT_(a, b)

void DGE_CAT(dge_launch_train_dch, DGE_DCH)(const TrainParams& p, int pol, bool big, unsigned blocks, unsigned threads, size_t shmem, hipStream_t st) {
    launch_train<DGE_DCH>(p, pol, big, blocks, threads, shmem, st);
}
