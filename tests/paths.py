"""Which kernel path the library takes for a flat fp32 search (no bf16 first pass), as
rag_index_last_path reports it -- mirrors the dispatch of rag_amd/csrc/api.hip so that tests can
assert that a case exercises the kernel it is meant for."""


def flat_fp32_path(n, d, nq, k):
    ld = (d + 3) // 4 * 4
    if 1 <= nq <= 4 and 1024 <= n <= 32768 and 1 <= k <= 1024 and ld <= 3072:
        return "direct"
    if nq >= 128 and 1024 <= n <= 32768 and 1 <= k <= 1024 and ld % 32 == 0 and nq * n * 4 <= 2 << 30:
        return "gemm-scores"          # every score written by the matrix-core kernel, one selection workgroup per query
    if 5 <= nq <= 32 and k <= 32 and n >= 4096 and ld % 128 == 0:
        nj = 1 if nq <= 16 else 2
        # (two 2 KiB transposer images per wave where they fit beside the query block, else one)
        if ld * nj * 64 + 8 * 2048 + 8 * nj * 16 * 8 + 48 <= 160 * 1024:
            return "narrow"
    if nq >= 5 and k <= 64 and ld % 32 == 0 and n >= 1:
        return "gemm"
    big_shape = nq >= 1024 or (nq > 128 and n >= 1 << 20 and -(-nq // 256) * 256 == -(-nq // 128) * 128)
    big_shape = big_shape and not (32768 < n < 1 << 19)      # short corpora take the 128 x 128 shape
    if nq > 64 and n >= 1 << 20 and 64 < k <= 200 and ld % 32 == 0:
        return "gemm-seeded"          # the selecting kernel with thresholds seeded from the first segments: from 65 queries on 2^20+ rows
    if big_shape and nq >= 2048 and 64 < k <= 200 and ld % 32 == 0 and n >= 1:
        return "gemm"                 # the 256 x 256 shape carries up to 200 keys per query, from 2048 queries
    if nq >= 2 and 64 < k <= 1024 and ld % 32 == 0 and n >= 1024:
        return "gemm-segments"        # k beyond the selecting kernel: score matrix + selection per 32768-row segment, one merge
    return "scan"
