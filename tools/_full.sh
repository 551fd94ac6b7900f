bash tools/profile_r03.sh r03 stats layers overlap pmc_iter 2>&1 | tail -120
