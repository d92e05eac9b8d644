bash tools/dbg/r3_final_b2.sh; bash tools/dbg/r3_final_c.sh
