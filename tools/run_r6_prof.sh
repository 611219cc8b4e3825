mkdir -p gpurun_out
V=$PWD/torchpq_amd/variants
P="python tools/scan_phase_profile.py --fused"
{
export TPQ_AMD_LIB=$V/libtorchpq_amd_prof.so
$P --m 64 --ds 2 --n-cells 16384 --cell 61 --n-probe 32
$P --m 64 --ds 2 --n-cells 4096 --cell 244 --n-probe 32
$P --m 32 --ds 4 --n-cells 4096 --cell 244 --n-probe 32
unset TPQ_AMD_LIB
} > gpurun_out/phase3.log 2>&1
cat gpurun_out/phase3.log
bash tools/kstats.sh ks_m32b python $PWD/tools/dump_route_check.py --one 32,4,4096,244,32,100,10000 | head -4
timeout 600 python tools/reference_grid.py --m 32 --repeats 1 --no-check --out gpurun_out/grid_m32b.json 2>&1 | tail -n 1
