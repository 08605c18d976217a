import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bionumpy_amd as bnp
gold = os.path.join(ROOT, "tests", "golden")
genome = bnp.open(os.path.join(gold, "sacCer3.fa.gz")).read()
seqs = bnp.change_encoding(genome.sequence, bnp.DNAEncoding)
for rep in range(6):
    index = bnp.KmerIndex.create_index(seqs, k=31)
torch.cuda.synchronize()
