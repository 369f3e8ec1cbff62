#!/bin/bash
bash tools/r2_exp.sh "ASVD_DUP2=1" "ASVD_DUP2=3" "ASVD_DUP2=7" "ASVD_SPARSE_FRAC=0.7"
