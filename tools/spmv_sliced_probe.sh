#!/bin/bash
# A/B of the plain and the XCD-sliced SpMV on the bench matrices (diagnostic).  usage: tools/spmv_sliced_probe.sh rmat|powerlaw|banded
which=${1:-rmat}
export PROBE_VARIANTS='[{"SPAM_SPMV_SLICED":0},{"SPAM_SPMV_SLICED":2,"SPAM_SPMV_LDS_KB":0},{"SPAM_SPMV_SLICED":2,"SPAM_SPMV_LDS_KB":64},{"SPAM_SPMV_SLICED":2,"SPAM_SPMV_LDS_KB":96},{"SPAM_SPMV_SLICED":2,"SPAM_SPMV_LDS_KB":96,"SPAM_SPMV_SLICED_TPB":1024},{"SPAM_SPMV_SLICED":2,"SPAM_SPMV_LDS_KB":0,"SPAM_SPMV_SLICED_TPB":512},{"SPAM_SPMV_SLICED":1}]'
python tools/spmv_probe.py $which 3
