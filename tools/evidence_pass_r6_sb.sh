R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6ev2; mkdir -p $O; cd $R
prof() {
  BENCH_ARGS="--streams $3 --steps 60 --blocks 5" timeout 500 tools/profile.sh $1 $2 > $O/prof_$1.log 2>&1
  cd $R; P=$R/gpurun_out/prof_$1; mkdir -p $O/$1
  cp $P/summary.txt $O/$1/summary.txt
  find $P -name "*kernel_stats.csv" -exec cp {} $O/$1/kernel_stats.csv \;
  rm -rf $P
}
prof r6_lisennet_4096 lisennet 4096
prof r6_bsrnn_t_4096 bsrnn_t 4096
prof r6_bsrnn_s_4096 bsrnn_s 4096
for t in r6_lisennet_4096 r6_bsrnn_t_4096 r6_bsrnn_s_4096; do echo "=== $t"; cat $O/$t/summary.txt | cut -c1-180 | head -80; done
