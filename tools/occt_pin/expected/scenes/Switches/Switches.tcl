variable Root [file dirname [file normalize [info script]]]

# Restore exported meshes
rtmeshread $Root/meshes/Mesh0.ply Mesh0 -group 
rtmeshread $Root/meshes/Mesh1.ply Mesh1 -group 
rtmeshread $Root/meshes/Mesh2.ply Mesh2 -group 
rtmeshread $Root/meshes/Mesh3.ply Mesh3 -group 
rtmeshread $Root/meshes/Mesh4.ply Mesh4 -group 
rtmeshread $Root/meshes/Mesh5.ply Mesh5 -group 
rtmeshread $Root/meshes/Mesh6.ply Mesh6 -group 

# Setup object 'Mesh0'
vdisplay Mesh0 -noupdate
vbsdf Mesh0 -Kc 0.0 0.0 0.0 -noupdate
vbsdf Mesh0 -Kd 1.0 0.30000001192092896 0.30000001192092896 -noupdate
vbsdf Mesh0 -Ks 0.0 0.0 0.0 -noupdate
vbsdf Mesh0 -Kt 0.0 0.0 0.0 -noupdate
vbsdf Mesh0 -baseRoughness 0.0 -noupdate
vbsdf Mesh0 -coatRoughness 0.0 -noupdate
vbsdf Mesh0 -Le 0.0 0.0 0.0 -noupdate
vbsdf Mesh0 -absorpColor 0.0 0.0 0.0 -noupdate
vbsdf Mesh0 -absorpCoeff 0.0 -noupdate
vbsdf Mesh0 -coatFresnel Constant 0.0 -noupdate
vbsdf Mesh0 -baseFresnel Constant 1.0 -noupdate

# Setup object 'Mesh1'
vdisplay Mesh1 -noupdate
vbsdf Mesh1 -Kc 0.0 0.0 0.0 -noupdate
vbsdf Mesh1 -Kd 0.30000001192092896 0.5 1.0 -noupdate
vbsdf Mesh1 -Ks 0.0 0.0 0.0 -noupdate
vbsdf Mesh1 -Kt 0.0 0.0 0.0 -noupdate
vbsdf Mesh1 -baseRoughness 0.0 -noupdate
vbsdf Mesh1 -coatRoughness 0.0 -noupdate
vbsdf Mesh1 -Le 0.0 0.0 0.0 -noupdate
vbsdf Mesh1 -absorpColor 0.0 0.0 0.0 -noupdate
vbsdf Mesh1 -absorpCoeff 0.0 -noupdate
vbsdf Mesh1 -coatFresnel Constant 0.0 -noupdate
vbsdf Mesh1 -baseFresnel Constant 1.0 -noupdate

# Setup object 'Mesh2'
vdisplay Mesh2 -noupdate
vbsdf Mesh2 -Kc 0.6000000238418579 0.6000000238418579 0.6000000238418579 -noupdate
vbsdf Mesh2 -Kd 0.4000000059604645 0.4000000059604645 0.4000000059604645 -noupdate
vbsdf Mesh2 -Ks 0.5 0.5 0.5 -noupdate
vbsdf Mesh2 -Kt 0.0 0.0 0.0 -noupdate
vbsdf Mesh2 -baseRoughness 0.15000000596046448 -noupdate
vbsdf Mesh2 -coatRoughness 0.25 -noupdate
vbsdf Mesh2 -Le 0.0 0.0 0.0 -noupdate
vbsdf Mesh2 -absorpColor 0.0 0.0 0.0 -noupdate
vbsdf Mesh2 -absorpCoeff 0.0 -noupdate
vbsdf Mesh2 -coatFresnel Dielectric 1.5 -noupdate
vbsdf Mesh2 -baseFresnel Schlick 0.800000011920929 0.800000011920929 0.800000011920929 -noupdate
rtmodel -sync default
rttexture Mesh2 "$Root/textures/tex0.png"

# Setup object 'Mesh3'
vdisplay Mesh3 -noupdate
vbsdf Mesh3 -Kc 0.0 0.0 0.0 -noupdate
vbsdf Mesh3 -Kd 0.7692307829856873 0.6153846383094788 0.1538461595773697 -noupdate
vbsdf Mesh3 -Ks 0.23076924681663513 0.23076924681663513 0.23076924681663513 -noupdate
vbsdf Mesh3 -Kt 0.0 0.0 0.0 -noupdate
vbsdf Mesh3 -baseRoughness 0.10000000149011612 -noupdate
vbsdf Mesh3 -coatRoughness 0.0 -noupdate
vbsdf Mesh3 -Le 0.0 0.0 0.0 -noupdate
vbsdf Mesh3 -absorpColor 0.0 0.0 0.0 -noupdate
vbsdf Mesh3 -absorpCoeff 0.0 -noupdate
vbsdf Mesh3 -coatFresnel Constant 0.0 -noupdate
vbsdf Mesh3 -baseFresnel Schlick 0.800000011920929 0.800000011920929 0.800000011920929 -noupdate

# Setup object 'Mesh4'
vdisplay Mesh4 -noupdate
vbsdf Mesh4 -Kc 0.30000001192092896 0.30000001192092896 0.30000001192092896 -noupdate
vbsdf Mesh4 -Kd 0.20000000298023224 0.20000000298023224 0.20000000298023224 -noupdate
vbsdf Mesh4 -Ks 0.0 0.0 0.0 -noupdate
vbsdf Mesh4 -Kt 0.699999988079071 0.699999988079071 0.699999988079071 -noupdate
vbsdf Mesh4 -baseRoughness 0.0 -noupdate
vbsdf Mesh4 -coatRoughness 0.20000000298023224 -noupdate
vbsdf Mesh4 -Le 0.0 0.0 0.0 -noupdate
vbsdf Mesh4 -absorpColor 0.0 0.0 0.0 -noupdate
vbsdf Mesh4 -absorpCoeff 0.0 -noupdate
vbsdf Mesh4 -coatFresnel Schlick 0.10000000149011612 0.10000000149011612 0.10000000149011612 -noupdate
vbsdf Mesh4 -baseFresnel Constant 1.0 -noupdate

# Setup object 'Mesh5'
vdisplay Mesh5 -noupdate
vbsdf Mesh5 -Kc 1.0 1.0 1.0 -noupdate
vbsdf Mesh5 -Kd 0.0 0.0 0.0 -noupdate
vbsdf Mesh5 -Ks 0.0 0.0 0.0 -noupdate
vbsdf Mesh5 -Kt 1.0 1.0 1.0 -noupdate
vbsdf Mesh5 -baseRoughness 0.0 -noupdate
vbsdf Mesh5 -coatRoughness 0.0 -noupdate
vbsdf Mesh5 -Le 0.0 0.0 0.0 -noupdate
vbsdf Mesh5 -absorpColor 0.800000011920929 0.800000011920929 1.0 -noupdate
vbsdf Mesh5 -absorpCoeff 6.0 -noupdate
vbsdf Mesh5 -coatFresnel Dielectric 1.5 -noupdate
vbsdf Mesh5 -baseFresnel Constant 1.0 -noupdate

# Setup object 'Mesh6'
vdisplay Mesh6 -noupdate
vbsdf Mesh6 -Kc 0.0 0.0 0.0 -noupdate
vbsdf Mesh6 -Kd 0.4166666567325592 0.7499999403953552 0.25 -noupdate
vbsdf Mesh6 -Ks 0.25 0.25 0.25 -noupdate
vbsdf Mesh6 -Kt 0.0 0.0 0.0 -noupdate
vbsdf Mesh6 -baseRoughness 0.0 -noupdate
vbsdf Mesh6 -coatRoughness 0.0 -noupdate
vbsdf Mesh6 -Le 0.0 0.0 0.0 -noupdate
vbsdf Mesh6 -absorpColor 0.0 0.0 0.0 -noupdate
vbsdf Mesh6 -absorpCoeff 0.0 -noupdate
vbsdf Mesh6 -coatFresnel Constant 0.0 -noupdate
vbsdf Mesh6 -baseFresnel Constant 1.0 -noupdate

# Restore scene hierarchy
rtmodel -sync default

# Restore view parameters
vcamera -perspective -fovy 45.0
vcamera -distance 1.0
vviewparams -proj -0.0 -1.0 -0.0
vviewparams -up 0.0 0.0 1.0
vviewparams -at 0.5 -0.19999999999999996 0.5
vviewparams -eye 0.5 -1.2 0.5
vviewparams -size 2.0

# Restore light source parameters
vlight clear
vlight add positional position 0.5 0.5 0.85 smoothness 0.12 intensity 12.0
rtlight 0 -color 1.0 1.0 1.0
vlight add directional direction -0.2 0.4 -1.0 smoothness 0.2 intensity 2.0
rtlight 1 -color 1.0 1.0 1.0
vtextureenv on $Root/textures/env.png
vrenderparams -ray -gi -rayDepth 6
