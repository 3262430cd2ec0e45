module raytracer_amd/native

go 1.16
