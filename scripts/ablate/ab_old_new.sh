cp scda_amd/libscda_ops.so /tmp/new.so
for rep in 1 2; do
python scripts/bench_conv_layers.py 2>/dev/null | grep -E "dec_res|conv5_x|dec_up1|conv3_2" | sed 's/^/NEW /'
cp scripts/ablate/libscda_ops_old.so scda_amd/libscda_ops.so
python scripts/bench_conv_layers.py 2>/dev/null | grep -E "dec_res|conv5_x|dec_up1|conv3_2" | sed 's/^/OLD /'
cp /tmp/new.so scda_amd/libscda_ops.so
done
