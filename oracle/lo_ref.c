/* lo_ref.c -- packed reference access (oracle; see lo.h).  pac2fa_core, src/bntseq.c:465-477. */
#include "lo_read.h"

int lo_pac_fetch(const lo_ref *R, int chr, int64_t start0, int32_t *len, uint8_t *dst)
{
    const int32_t clen = R->seq_len[chr - 1];
    if (start0 > clen || start0 < 0) return -1;                      /* the reference exit(1)s here, :469-472 */
    if (start0 + *len > clen) *len = (int32_t)(clen - start0);       /* silently shortened at the contig end, :474 */
    int64_t k = R->seq_offset[chr - 1] + start0;
    for (int32_t i = 0; i < *len; ++i, ++k) dst[i] = R->pac[k >> 2] >> ((~k & 3) << 1) & 3;   /* _get_pac, :242 */
    return 0;
}
