bash tools/profile_b4.sh r03_f && echo profile done && bash tools/pmc_collect.sh && echo pmc done
