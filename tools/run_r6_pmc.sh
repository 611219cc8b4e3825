mkdir -p gpurun_out
G="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for shape in 64,2,16384,61,32,100,10000 64,2,1024,977,32,100,10000 32,4,4096,244,32,100,10000; do
  tag=pmc_$(echo $shape | tr ',' '_')
  echo "== $shape"
  bash tools/pmc.sh $tag "scan_packed_kernel" "$G" python $PWD/tools/dump_route_check.py --one $shape --iters 3
  bash tools/kstats.sh ${tag}_ks python $PWD/tools/dump_route_check.py --one $shape --iters 3 | head -3
done
