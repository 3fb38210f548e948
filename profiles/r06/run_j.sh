# pipeline test of the PMC collection (pmc passes only; the bench trace comes with the final tree)
bash profiles/pmc_r06.sh pmc-only 2>&1 | tail -30
ls -la gpurun_out/prof_r06/
