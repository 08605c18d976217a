cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fastq or fasta" > gpurun_out/t_fq.log 2>&1
tail -n 3 gpurun_out/t_fq.log
for i in 1 2; do timeout 300 python scripts/exp/exp_fq.py 20000000 150 1 5 | grep encoder | cut -c 1-200; done
timeout 300 python scripts/exp/exp_fq.py 20000000 100 1 5 | grep encoder | cut -c 1-200
timeout 300 python scripts/exp/exp_fq.py 20000000 50 1 5 | grep encoder | cut -c 1-200
timeout 300 python scripts/exp/exp_fq.py 1000000 5000 1 5 | grep encoder | cut -c 1-200
