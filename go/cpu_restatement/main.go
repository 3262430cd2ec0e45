// cpu_restatement -- a stdlib-only Go program that keeps the data layout and
// control flow of the reference's CPU ray caster (pointer tree of BVH nodes,
// []*Triangle, a heap-allocated ray and result per cast, recursive left-first
// traversal; src/backend/models/{ray,aabb,triangle,bvh}.go and
// process/trace.go:121-152) so that "Go on the CPU" can be timed on the host
// next to the GPU path.  It is NOT the reference (which only builds for
// js/wasm and depends on mathgl, gwob and gonum) and it was NOT compiled in the
// environment this repository was developed in: that environment has no Go
// toolchain.  bench.py runs it only when `go version` works.
//
//	go run . -tris 100000 -rays 2000000
//
// Scene: triangle soup from a splitmix64 stream (centroids U([-1,1]^3), vertex
// offsets U([-0.02,0.02]^3)), SAH tree with leaves of at most 6 triangles.
// Rays: a 16:9 pinhole camera at (0,0,4) looking down -Z, fov 45 degrees.
// Prints one JSON line: single-goroutine and all-core Mrays/s.
package main

import (
	"flag"
	"fmt"
	"math"
	"runtime"
	"sort"
	"sync"
	"time"
)

type Vec3 [3]float32

func (a Vec3) Sub(b Vec3) Vec3     { return Vec3{a[0] - b[0], a[1] - b[1], a[2] - b[2]} }
func (a Vec3) Add(b Vec3) Vec3     { return Vec3{a[0] + b[0], a[1] + b[1], a[2] + b[2]} }
func (a Vec3) Mul(s float32) Vec3  { return Vec3{a[0] * s, a[1] * s, a[2] * s} }
func (a Vec3) Dot(b Vec3) float32  { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2] }
func (a Vec3) Cross(b Vec3) Vec3 {
	return Vec3{a[1]*b[2] - a[2]*b[1], a[2]*b[0] - a[0]*b[2], a[0]*b[1] - a[1]*b[0]}
}
func (a Vec3) Len() float32 { return float32(math.Sqrt(float64(a.Dot(a)))) }
func (a Vec3) Normalize() Vec3 {
	l := 1.0 / a.Len()
	return a.Mul(l)
}

// models/ray.go:9-42
type Ray struct {
	Origin, Direction, InvDirection Vec3
	Sign                            [3]int
}

func NewRay(origin, direction Vec3) *Ray {
	inv := Vec3{1.0 / direction[0], 1.0 / direction[1], 1.0 / direction[2]}
	r := &Ray{Origin: origin, Direction: direction, InvDirection: inv}
	for k := 0; k < 3; k++ {
		if inv[k] < 0 {
			r.Sign[k] = 1
		}
	}
	return r
}

// models/aabb.go:11-22, 62-99
type AABB struct {
	Bounds [2]Vec3
}

func (box *AABB) RayIntersect(ray *Ray) (bool, float32, float32) {
	tmin := (box.Bounds[ray.Sign[0]][0] - ray.Origin[0]) * ray.InvDirection[0]
	tmax := (box.Bounds[1-ray.Sign[0]][0] - ray.Origin[0]) * ray.InvDirection[0]
	tymin := (box.Bounds[ray.Sign[1]][1] - ray.Origin[1]) * ray.InvDirection[1]
	tymax := (box.Bounds[1-ray.Sign[1]][1] - ray.Origin[1]) * ray.InvDirection[1]
	if tmin > tymax || tymin > tmax {
		return false, 0, 0
	}
	if tymin > tmin {
		tmin = tymin
	}
	if tymax < tmax {
		tmax = tymax
	}
	tzmin := (box.Bounds[ray.Sign[2]][2] - ray.Origin[2]) * ray.InvDirection[2]
	tzmax := (box.Bounds[1-ray.Sign[2]][2] - ray.Origin[2]) * ray.InvDirection[2]
	if tmin > tzmax || tzmin > tmax {
		return false, 0, 0
	}
	if tzmin > tmin {
		tmin = tzmin
	}
	if tzmax < tmax {
		tmax = tzmax
	}
	return true, tmin, tmax
}

// models/triangle.go:12-31, 45-65, 78-103
type Triangle struct {
	Vertices            [3]Vec3
	Normal              Vec3
	Edge0, Edge1, Edge2 Vec3
	Index               int
}

func NewTriangle(v0, v1, v2 Vec3, index int) *Triangle {
	return &Triangle{
		Vertices: [3]Vec3{v0, v1, v2},
		Normal:   v1.Sub(v0).Cross(v2.Sub(v0)).Normalize(),
		Edge0:    v1.Sub(v0), Edge1: v2.Sub(v1), Edge2: v0.Sub(v2),
		Index: index,
	}
}

func (t *Triangle) Center() Vec3 {
	return t.Vertices[0].Add(t.Vertices[1]).Add(t.Vertices[2]).Mul(1.0 / 3.0)
}

func (t *Triangle) RayIntersect(ray *Ray) (float32, float32, float32) {
	v0v2 := t.Edge2.Mul(-1)
	pvec := ray.Direction.Cross(v0v2)
	det := t.Edge0.Dot(pvec)
	if det < 0.0001 {
		return -1, 0, 0
	}
	invdet := 1.0 / det
	tvec := ray.Origin.Sub(t.Vertices[0])
	u := tvec.Dot(pvec) * invdet
	if u < 0 || u > 1 {
		return -1, 0, 0
	}
	qvec := tvec.Cross(t.Edge0)
	v := ray.Direction.Dot(qvec) * invdet
	if v < 0 || u+v > 1 {
		return -1, 0, 0
	}
	return v0v2.Dot(qvec) * invdet, u, v
}

// models/bvh.go:13-29
type BVHNode struct {
	LeftChild, RightChild *BVHNode
	Bounds                *AABB
	StartIndex, EndIndex  int
}

func bounds(tris []*Triangle, start, end int) *AABB {
	mn := Vec3{math.MaxFloat32, math.MaxFloat32, math.MaxFloat32}
	mx := Vec3{-math.MaxFloat32, -math.MaxFloat32, -math.MaxFloat32}
	for i := start; i <= end; i++ {
		for _, v := range tris[i].Vertices {
			for k := 0; k < 3; k++ {
				if v[k] < mn[k] {
					mn[k] = v[k]
				}
				if v[k] > mx[k] {
					mx[k] = v[k]
				}
			}
		}
	}
	return &AABB{Bounds: [2]Vec3{mn, mx}}
}

func area(b *AABB) float32 {
	d := b.Bounds[1].Sub(b.Bounds[0])
	return 2 * (d[0]*d[1] + d[0]*d[2] + d[1]*d[2])
}

func grow(b *AABB, t *Triangle) {
	for _, v := range t.Vertices {
		for k := 0; k < 3; k++ {
			if v[k] < b.Bounds[0][k] {
				b.Bounds[0][k] = v[k]
			}
			if v[k] > b.Bounds[1][k] {
				b.Bounds[1][k] = v[k]
			}
		}
	}
}

func emptyBox() *AABB {
	return &AABB{Bounds: [2]Vec3{
		{math.MaxFloat32, math.MaxFloat32, math.MaxFloat32},
		{-math.MaxFloat32, -math.MaxFloat32, -math.MaxFloat32}}}
}

func sortAxis(tris []*Triangle, start, end, axis int) {
	part := tris[start : end+1]
	sort.Slice(part, func(i, j int) bool {
		a, b := part[i].Center()[axis], part[j].Center()[axis]
		if a == b {
			return part[i].Index < part[j].Index
		}
		return a < b
	})
}

// bvh.go:56-114, 146-212: full-sweep SAH over three axes, sorting in place
func build(tris []*Triangle, start, end, depth, maxLeaf, maxDepth int) *BVHNode {
	node := &BVHNode{Bounds: bounds(tris, start, end), StartIndex: start, EndIndex: end}
	count := end - start + 1
	if count <= maxLeaf || depth >= maxDepth {
		return node
	}
	bestCost, bestAxis, bestSplit := float32(math.MaxFloat32), -1, -1
	left := make([]float32, count)
	for axis := 0; axis < 3; axis++ {
		sortAxis(tris, start, end, axis)
		box := emptyBox()
		for i := start; i <= end; i++ {
			grow(box, tris[i])
			left[i-start] = area(box)
		}
		box = emptyBox()
		for i := end; i > start; i-- {
			grow(box, tris[i])
			cost := left[i-1-start]*float32(i-start) + area(box)*float32(end-i+1)
			if cost < bestCost {
				bestCost, bestAxis, bestSplit = cost, axis, i
			}
		}
	}
	if bestAxis < 0 {
		bestAxis, bestSplit = 0, start+count/2
	}
	sortAxis(tris, start, end, bestAxis)
	node.LeftChild = build(tris, start, bestSplit-1, depth+1, maxLeaf, maxDepth)
	node.RightChild = build(tris, bestSplit, end, depth+1, maxLeaf, maxDepth)
	return node
}

// bvh.go:117-144
func (node *BVHNode) WalkNode(tris []*Triangle, ray *Ray, tmin, umin, vmin *float32, hit **Triangle) {
	if node.LeftChild == nil {
		for i := node.StartIndex; i <= node.EndIndex; i++ {
			tri := tris[i]
			if tri.Normal.Dot(ray.Direction) > 0 {
				continue
			}
			t, u, v := tri.RayIntersect(ray)
			if t > 0 && t < *tmin {
				*tmin, *umin, *vmin, *hit = t, u, v, tri
			}
		}
		return
	}
	if ok, t0, t1 := node.LeftChild.Bounds.RayIntersect(ray); ok && t0 < *tmin && t1 > 0 {
		node.LeftChild.WalkNode(tris, ray, tmin, umin, vmin, hit)
	}
	if ok, t0, t1 := node.RightChild.Bounds.RayIntersect(ray); ok && t0 < *tmin && t1 > 0 {
		node.RightChild.WalkNode(tris, ray, tmin, umin, vmin, hit)
	}
}

// process/trace.go:11-17, 121-152
type RaycastResult struct {
	Triangle *Triangle
	T, U, V  float32
	Point    Vec3
}

func rayCast(root *BVHNode, tris []*Triangle, ray *Ray, initialTmin float32) *RaycastResult {
	tmin := initialTmin
	var u, v float32
	var tri *Triangle
	root.WalkNode(tris, ray, &tmin, &u, &v, &tri)
	if tmin < math.MaxFloat32 {
		return &RaycastResult{Triangle: tri, T: tmin, U: u, V: v, Point: ray.Origin.Add(ray.Direction.Mul(tmin))}
	}
	return nil
}

type splitmix struct{ x uint64 }

func (s *splitmix) unit() float64 {
	s.x += 0x9E3779B97F4A7C15
	z := s.x
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9
	z = (z ^ (z >> 27)) * 0x94D049BB133111EB
	z ^= z >> 31
	return float64(z>>40) * (1.0 / 16777216.0)
}

func castRange(root *BVHNode, tris []*Triangle, first, last, width, height int) int {
	hits := 0
	tanHalf := math.Tan(45.0 / 2 * math.Pi / 180)
	aspect := float64(width) / float64(height)
	for i := first; i < last; i++ {
		p := i % (width * height)
		x, y := p%width, p/width
		px := (2*(float64(x)+0.5)/float64(width) - 1) * tanHalf * aspect
		py := (1 - 2*(float64(y)+0.5)/float64(height)) * tanHalf
		dir := Vec3{float32(px), float32(py), -1}.Normalize()
		ray := NewRay(Vec3{0, 0, 4}, dir) // one heap ray per cast, as in the reference
		if res := rayCast(root, tris, ray, math.MaxFloat32); res != nil && res.Triangle != nil {
			hits++
		}
	}
	return hits
}

func main() {
	nTris := flag.Int("tris", 100000, "triangles of the synthetic soup")
	nRays := flag.Int("rays", 2000000, "rays to cast per timing leg")
	flag.Parse()
	rng := &splitmix{x: 0x5EED0001}
	tris := make([]*Triangle, 0, *nTris)
	for i := 0; i < *nTris; i++ {
		var c Vec3
		for k := 0; k < 3; k++ {
			c[k] = float32(rng.unit()*2 - 1)
		}
		var v [3]Vec3
		for j := 0; j < 3; j++ {
			for k := 0; k < 3; k++ {
				v[j][k] = c[k] + float32((rng.unit()*2-1)*0.02)
			}
		}
		tris = append(tris, NewTriangle(v[0], v[1], v[2], i))
	}
	t0 := time.Now()
	root := build(tris, 0, len(tris)-1, 0, 6, 32)
	buildS := time.Since(t0).Seconds()
	width, height := 1920, 1080

	t0 = time.Now()
	hits1 := castRange(root, tris, 0, *nRays, width, height)
	one := time.Since(t0).Seconds()

	workers := runtime.NumCPU()
	var wg sync.WaitGroup
	total := make([]int, workers)
	t0 = time.Now()
	for w := 0; w < workers; w++ {
		wg.Add(1)
		go func(w int) {
			defer wg.Done()
			per := (*nRays + workers - 1) / workers
			first, last := w*per, (w+1)*per
			if last > *nRays {
				last = *nRays
			}
			if first < last {
				total[w] = castRange(root, tris, first, last, width, height)
			}
		}(w)
	}
	wg.Wait()
	all := time.Since(t0).Seconds()
	hitsAll := 0
	for _, h := range total {
		hitsAll += h
	}
	fmt.Printf("{\"kind\": \"go restatement (stdlib only, reference data layout)\", \"go\": %q, \"triangles\": %d, "+
		"\"rays\": %d, \"build_s\": %.2f, \"single_goroutine_mrays_per_s\": %.3f, \"all_cores_mrays_per_s\": %.3f, "+
		"\"cores\": %d, \"hits\": %d, \"hits_all_cores\": %d}\n",
		runtime.Version(), *nTris, *nRays, buildS, float64(*nRays)/one/1e6, float64(*nRays)/all/1e6, workers, hits1, hitsAll)
}
