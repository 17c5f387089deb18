import json, os, sys
sys.path.insert(0, "/root/repo")
import bench
for case in [("2b scripted 32768", "OscPanScripted", 32768), ("2b scripted 65536", "OscPanScripted", 65536),
             ("2e env per voice 65536", "OscPanEnvScripted", 65536), ("3b scripted filter 65536", "OscFilterPanScripted", 65536)]:
    r = bench.engine_in_loop(cases=[case])
    print(json.dumps({"case": case[0], **r["cases"][case[0]]}), flush=True)
