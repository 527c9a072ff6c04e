for rpb in 64 128 256 512; do echo "== RPB=$rpb"; SAME_DENSE_RPB=$rpb python tools/dense_probe.py 100000 3,8,12,16,20; done
echo "== f32"; for rpb in 128 512; do echo "== RPB=$rpb"; SAME_DENSE_RPB=$rpb python tools/dense_probe.py 100000 3,8,20 f32; done
