cd tools/ubench
for a in "64 64 160 8 0 8 0 128" "64 64 160 8 1 8 0 128" "64 64 160 16 0 8 0 128" "128 128 160 4 0 8 0 64" "128 128 160 4 1 8 0 64" "128 128 160 8 0 8 0 64"; do
  for b in conv_sp_probe_old conv_sp_probe conv_sp_probe_old conv_sp_probe; do echo "### $b $a: $(timeout 60 ./$b $a | grep -E 'without|PARITY' | cut -c1-60 | tr '\n' ' ')"; done
done
