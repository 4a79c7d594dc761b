ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fused; mkdir -p $OUT
fused() { tag=$1; shift
  env "$@" timeout 90 python $ROOT/bench.py --steps 400 --warmup 20 --no-cpu-baseline > $OUT/f_$tag.json 2> $OUT/f_$tag.err
  python -c "import json; j=json.load(open('$OUT/f_$tag.json')); print('fused $tag', round(j['value']/1e12,3), 'T evals/s', round(j['ms_per_step']*1e3,1), 'us/step kernel', round(j['roofline']['kernel_ms']*1e3,1))" || tail -3 $OUT/f_$tag.err
}
fused cs2 NHDFIT_CHOOSE_SPLIT=2
fused cs4 NHDFIT_CHOOSE_SPLIT=4
fused cs8 NHDFIT_CHOOSE_SPLIT=8
fused cs8t640 NHDFIT_CHOOSE_SPLIT=8 NHDFIT_FIT_BLOCKS=640
fused cs8t1024 NHDFIT_CHOOSE_SPLIT=8 NHDFIT_FIT_BLOCKS=1024
fused cs8p0 NHDFIT_CHOOSE_SPLIT=8 NHDFIT_SIDE_PRIO=0
