cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fastq or fasta" > gpurun_out/t_fq.log 2>&1
tail -n 5 gpurun_out/t_fq.log
FQ_ARGS="20000000 150 1 2" bash scripts/exp/run10.sh
timeout 300 python scripts/exp/exp_fq.py 20000000 100 1 3 | grep encoder
timeout 300 python scripts/exp/exp_fq.py 20000000 50 1,0 3 | grep encoder
