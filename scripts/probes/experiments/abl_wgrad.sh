# Ablation of the wgrad kernel phases: BRATS_WGRAD_DEBUG bits 1 loads dropped, 2 no MFMA, 4 no tiles, 8 no LDS writes (tap-plane), 16 no load instructions (tap-plane)
for d in ${ABL:-0 1 2 3 8 10 11}; do echo "debug=$d"; BRATS_WGRAD_DEBUG=$d python scripts/time_conv.py ${SHAPE:-48 48 128} 1 10 2>&1 | grep wgrad; done
