"""Constants of the OPT-IN GPU entropy stage's host interface (csrc/deflate.hip, include/aej.h): the sizes of the symbol histogram the
kernels count and of the per-layer Huffman tables the library's host helper ``aej_deflate_build_tables`` builds from it.  (The readable
restatement of that construction is test infrastructure: tests/deflate_reference.py.)"""
HIST_BINS = 320          # AEJ_DEFLATE_HIST_BINS: 286 literal / length symbols, then 30 distance symbols
TABLE_WORDS = 448        # AEJ_DEFLATE_TABLE_WORDS
N_LITLEN, N_DIST = 286, 30
