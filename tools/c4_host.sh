for a in "--steps 5 --width 5760 --height 3240 --frames 8" "--steps 5 --path host --width 5760 --height 3240 --frames 8" "--steps 3 --path host --width 5760 --height 3240 --frames 64" "--steps 5 --path host --frames 32"; do
 echo "# $a"; python bench.py --no-cpu-baseline --no-e2e --no-refbytes --no-lanes $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('pcie'))"
done
