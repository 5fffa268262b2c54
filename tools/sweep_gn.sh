for cfg in "2 2" "4 1" "4 2" "4 4" "8 2" "8 4"; do set -- $cfg; echo "ppt $1 group $2"; python tools/probe_gn.py --batch 64 --level 3 --launches 20 --sigma 0.5 --ppt $1 --group $2 | tail -1; done
