for d in ${ABL:-0 1 2 3 8 10 11}; do echo "debug=$d"; BRATS_WGRAD_DEBUG=$d python scripts/time_conv.py ${SHAPE:-48 48 128} 1 10 2>&1 | grep wgrad; done
