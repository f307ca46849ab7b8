/* tools/kindel_tools.h -- C-ABI of tools/libkindel_tools.so: test / bench tools that are NOT part of the product library
 * (include/kindel_hip.h is the product boundary). */
#ifndef KINDEL_TOOLS_H
#define KINDEL_TOOLS_H
#include <stdint.h>

#include "../include/kindel_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Write a host batch as a BGZF-compressed BAM (synthetic inputs for end-to-end runs; parallel deflate).  sort_order: the @HD SO
 * value; n_threads 0 = the host's share; level: zlib's.  KD_WRITE_BAM_QUAL=phred in the environment: Phred-like qualities (the
 * file then compresses like sequencer output, 2 - 4 x) instead of absent ones (15 x).  Returns KD_OK or a KD_E_* code. */
int kd_write_bam(const char *path, const kd_batch *host_batch, uint32_t n_contigs, const char *const *names,
                 const uint32_t *lens, const char *sort_order, int n_threads, int level);
const char *kd_tools_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
