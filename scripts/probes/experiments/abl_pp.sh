# Ablation of the ping-pong igemm kernel phases (BRATS_CONV_PP=1): BRATS_CONV_DEBUG bits 1 no LDS-DMA, 2 no MFMA, 4 no epilogue
BRATS_CONV_PP=1; export BRATS_CONV_PP
for d in ${ABL:-0 1 2 4 5 3 7}; do echo "debug=$d"; BRATS_CONV_DEBUG=$d python scripts/time_conv.py ${SHAPE:-48 96 128} 1 10 2>&1 | grep fwd; done
