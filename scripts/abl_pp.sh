for d in ${ABL:-0 1 2 4 5 3 7}; do echo "debug=$d"; BRATS_CONV_DEBUG=$d python scripts/time_conv.py ${SHAPE:-48 96 128} 1 10 2>&1 | grep fwd; done
