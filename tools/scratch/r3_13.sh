#!/bin/bash
cd $GRAFT_REPO_ROOT
export KB_ONLY=wgrad
E2E_WG_BF3=2 bash tools/pmc_mfma.sh pmc_bf3v2 L0_64x32 2>&1 | grep -A18 "bf3" | head -60
E2E_WG_BF3=1 bash tools/pmc_mfma.sh pmc_bf3v1 L0_64x32 2>&1 | grep -A18 "bf3" | head -40
E2E_WG_BF3=2 bash tools/pmc_one.sh pmc_bf3_traffic L0_64x32 2>&1 | grep -i "bf3" | head
