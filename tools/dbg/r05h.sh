cd tools/micro
for args in "8 64 256 512 64 0" "8 64 128 256 128 0" "8 128 128 256 128 0" "8 128 64 128 256 0" "8 256 64 128 256 1" "2 64 37 53 64 0" "2 64 37 53 128 1" "3 72 16 24 64 1"; do
  for ck in 8 4; do echo "== $args ck=$ck"; ./wino_micro $args 10 $ck 2>&1 | tail -2; done
done
