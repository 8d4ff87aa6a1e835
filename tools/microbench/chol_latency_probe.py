"""Per-row critical path of the Cholesky row kernel (one row, 16 entries) at several k."""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools/microbench')
import numpy as np
from chol_rows_probe import probe
for k,dt in ((257,np.float32),(129,np.float64)):
    for R,nnz in ((1,16),(2560,16)):
        t,ok=probe(R,nnz,k=k,dtype=dt,reps=10)
        print(k,dt.__name__,"R=%d nnz=%d: %.1f us  (per row-slot %.1f us)"%(R,nnz,t*1e6,t*1e6/max(1,R/256)),ok,flush=True)
