cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fastq or fasta" > gpurun_out/t_fq.log 2>&1
tail -n 15 gpurun_out/t_fq.log
timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed > gpurun_out/b_fq.log 2>&1
tail -n 3 gpurun_out/b_fq.log | cut -c 1-2500
