// What csrc/nrm_de_sparse.hip (the gathers) and csrc/nrm_design_lists.hip (the kernels that build its lists) agree on.
#pragma once

#define DS_CH 2048   // cells per chunk: 32 KB of records, three workgroups per CU.  Measured with the design rows dealt once for all chunks: 4096 cells (two
                     // workgroups per CU) 1.82 ms, 8192 (one) 2.11, 2048 1.88 (17 % more padded entries); dealt per chunk (sig) the lists of a wave are
                     // equally long whatever the chunk size and the waves in flight decide: 4096 cells 1.69 ms, 2048 cells 1.56.
#define DS_T 512     // threads per workgroup
#define DS_G 2       // design rows per thread: 1024 per pass over the expression matrix
#define DS_PASS (DS_T * DS_G)  // positions dealt together: the design rows are sorted by their entries inside every block of DS_PASS slots

// bits of info[2] of nrm_design_count (include/normalisr_hip.h): what the design's entries are like
#define DL_NOTONE NRM_DESIGN_NOTONE
#define DL_NEG NRM_DESIGN_NEG
#define DL_GT1 NRM_DESIGN_GT1
#define DL_HAS1 NRM_DESIGN_HAS1
#define DL_NAN NRM_DESIGN_NAN
