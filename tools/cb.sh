cd tools
for shape in "64 64 64 192 192" "64 64 64 384 192" "64 64 64 384 384" "64 32 32 576 576" "64 32 32 960 576" "64 16 16 768 768" "64 16 16 1344 768"; do
  for cfg in "96 1 2" "96 1 3" "128 1 2" "128 1 3" "192 1 2"; do
    set -- $cfg; bn=$1
    co=$(echo $shape | awk '{print $5}')
    if [ $((co % bn)) -ne 0 ]; then continue; fi
    timeout 60 ./conv_bench.out $shape 9 1 $cfg 1
  done
done
