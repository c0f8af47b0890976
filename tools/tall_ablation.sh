#!/bin/bash
# What is lkg_gemm_tall's k loop waiting for?  Ablation builds of the library (each leaves ONE ingredient out; the results
# are wrong by construction, only the time matters), to be run side by side on the GPU box:
#   tools/tall_ablation.sh build      (here: writes literalkg_amd/lib/abl_<name>.so, then rebuilds the real library)
#   tools/tall_ablation.sh run        (GPU box: times every one with tools/tall_variants_micro.py, restores the real library)
set -e
cd "$(dirname "$0")/.."
NAMES="NO_MFMA NO_STAGE NO_DMA NO_STORE"
if [ "$1" = build ]; then
    for n in $NAMES; do
        LKG_EXTRA_HIPCC_FLAGS="-DLKG_ABL_$n" python -m literalkg_amd.build >/dev/null 2>&1
        cp literalkg_amd/lib/liblkg_hip.so literalkg_amd/lib/abl_$n.so
    done
    python -m literalkg_amd.build >/dev/null 2>&1
else
    cp literalkg_amd/lib/liblkg_hip.so /tmp/lkg_real.so
    echo "== real"; python tools/tall_variants_micro.py --variants 256x1 --rounds 5 2>&1 | grep "^linear\|^gate f"
    for n in $NAMES; do
        cp literalkg_amd/lib/abl_$n.so literalkg_amd/lib/liblkg_hip.so
        echo "== $n"; python tools/tall_variants_micro.py --variants 256x1 --rounds 5 2>&1 | grep "^linear\|^gate f"
    done
    cp /tmp/lkg_real.so literalkg_amd/lib/liblkg_hip.so
fi
