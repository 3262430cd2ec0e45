module raytracer_amd/cpu_restatement

go 1.16
