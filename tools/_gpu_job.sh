O=gpurun_out/r06zn_solo.txt
python tools/rows_ab.py solo128 2>&1 | grep "^{" > $O
for v in 32 64 256 1024; do ETLG_LIB_PATH=$PWD/build/variants/solo$v.so python tools/rows_ab.py solo$v 2>&1 | grep "^{" >> $O; done
