# Material-preview workload for cadrays_amd.run_script -- this project's own script, parameterised like the preview
# recipe the reference application ships (128 x 128 target, ray depth 10, 8000 frames per material, a ball on a
# two-tone 12 x 12 tiled floor): python -m cadrays_amd.run_script tools/material_preview.tcl --outdir out
set stock_names {brass bronze copper gold pewter plaster plastic silver steel stone shiny_plastic satin metalized neon_gnc chrome aluminium obsidian neon_phc jade charcoal water glass diamond transparent}
set frames_per_material 8000

vinit name=Preview w=128 h=128
vcamera -persp
vviewparams -scale 18 -eye 44.49 -0.15 33.93 -at -14.20 -0.15 7.0 -up -0.48 0.00 0.88

# floor: 144 slabs of 10 x 10 x 0.1, alternating grey levels
box slab 10 10 0.1
for {set row 0} {$row < 12} {incr row} {
  for {set col 0} {$col < 12} {incr col} {
    set k [expr 12 * $row + $col]
    eval compound slab piece$k
    explode piece$k
    ttranslate piece${k}_1 [expr $row * 10 - 90] [expr $col * 10 - 60] -0.15
    vdisplay piece${k}_1
    if {($row + $col) % 2 == 1} { vbsdf piece${k}_1 -kd 0.85 } else { vbsdf piece${k}_1 -kd 0.45 }
  }
}

psphere probe 14
vdisplay probe
vsetlocation probe 0 0 14

vlight del 1
vlight change 0 head 0 direction -0.25 -1 -1 sm 0.3 int 10
vrenderparams -ray -gi -rayDepth 10

foreach name $stock_names {
  vsetmaterial probe $name
  vfps $frames_per_material
  vdump $name.png
}
