#!/bin/bash
# round 3, last trip: HBM traffic of the headline kernel at HEAD (separate --pmc passes, as the guide prescribes)
O=gpurun_out/r3pmc; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/run_fwd.py 10 > /dev/null 2>$O/err.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/run_fwd.py 10 > /dev/null 2>>$O/err.txt
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
find $O -name "*.db" -delete
