for m in 4 8 16; do for k in 1 100; do
echo "m=$m k=$k plain: $(python tools/scan_microbench.py --m $m --k $k --layouts packed --iters 20 2>/dev/null)"
echo "m=$m k=$k conflict-free: $(python tools/scan_microbench.py --m $m --k $k --layouts packed --iters 20 --conflict-free 2>/dev/null)"
done; done
