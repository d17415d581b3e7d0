Prev '\\n' table:
Table:
\\n 1 0
A 5 10100
C 4 1110
D 4 1101
E 6 111110
F 7 1111110
I 5 10110
M 5 10101
N 4 1001
O 7 1111111
P 4 1000
Q 6 101111
S 4 1100
U 6 101110
V 5 11110
Prev '\\sp' table:
Table:
A 7 0111110
C 7 1011011
D 6 011100
E 8 10111110
F 8 10111111
I 7 1011110
L 10 1111111110
M 7 1111110
N 6 011110
O 10 1111111111
P 6 101100
Q 8 11111010
S 6 011101
U 9 111111110
V 7 1011010
a 4 1110
b 7 1111100
c 5 11010
d 5 11110
e 3 000
f 5 11001
g 8 11111011
h 8 11111110
i 4 0101
j 7 0111111
l 4 0010
m 4 0011
n 4 1000
o 6 110111
p 4 0110
q 6 110110
r 6 101110
s 4 1010
t 4 0100
u 5 11000
v 4 1001
Prev ',' table:
Table:
\\sp 1 1
Prev '.' table:
Table:
\\n 1 0
\\sp 1 1
Prev ';' table:
Table:
\\sp 1 1
Prev 'A' table:
Table:
e 1 0
l 1 1
Prev 'C' table:
Table:
l 2 10
r 2 11
u 1 0
Prev 'D' table:
Table:
o 1 1
u 1 0
Prev 'E' table:
Table:
t 1 1
Prev 'F' table:
Table:
u 1 1
Prev 'I' table:
Table:
n 1 1
Prev 'L' table:
Table:
o 1 1
Prev 'M' table:
Table:
a 1 1
o 1 0
Prev 'N' table:
Table:
a 1 0
u 1 1
Prev 'O' table:
Table:
r 1 1
Prev 'P' table:
Table:
e 2 11
h 2 10
r 1 0
Prev 'Q' table:
Table:
u 1 1
Prev 'S' table:
Table:
e 1 1
u 1 0
Prev 'U' table:
Table:
t 1 1
Prev 'V' table:
Table:
e 1 1
i 1 0
Prev 'a' table:
Table:
\\sp 3 110
, 6 111110
. 5 01111
b 7 1111111
c 3 000
d 6 111010
e 4 0010
g 5 01110
l 4 0011
m 3 100
n 5 11110
o 7 1111110
p 5 11100
r 4 1010
s 4 0110
t 3 010
u 4 1011
v 7 1110111
x 7 1110110
Prev 'b' table:
Table:
e 3 110
h 4 1110
i 2 10
l 5 11111
o 5 11110
u 1 0
Prev 'c' table:
Table:
\\sp 2 00
, 6 111110
. 6 111111
c 5 11110
e 4 1110
i 2 01
o 3 101
t 3 100
u 3 110
Prev 'd' table:
Table:
\\sp 2 01
, 6 111111
. 6 111110
a 3 110
i 2 10
o 4 1110
r 5 11110
u 2 00
Prev 'e' table:
Table:
\\sp 3 001
, 7 1111100
. 6 111100
; 9 111111111
a 8 11111110
c 4 0111
d 5 11010
e 7 1101110
f 7 1101111
g 5 11000
h 7 1101101
i 7 1101100
l 3 010
m 4 0110
n 3 101
o 7 1111101
p 9 111111110
q 6 111101
r 3 000
s 4 1110
t 3 100
u 5 11001
x 7 1111110
Prev 'f' table:
Table:
a 2 00
e 2 01
f 3 111
i 2 10
r 3 110
Prev 'g' table:
Table:
\\sp 4 1110
e 2 01
i 2 10
n 3 110
r 4 1111
u 2 00
Prev 'h' table:
Table:
\\sp 3 110
, 4 1110
. 4 1111
a 2 00
e 3 010
i 2 10
o 3 011
Prev 'i' table:
Table:
\\sp 5 11010
, 7 1011110
. 7 1111110
a 5 11110
b 4 1110
c 4 1010
d 4 0110
e 5 11001
f 7 1011111
g 6 111110
l 6 110111
m 4 0111
n 3 010
o 7 1111111
p 5 10110
q 5 11000
s 2 00
t 3 100
u 6 110110
v 6 101110
Prev 'j' table:
Table:
u 1 1
Prev 'l' table:
Table:
\\sp 5 10111
, 7 1111111
. 7 1111110
a 2 00
e 3 110
i 2 01
l 3 100
o 4 1110
p 6 111110
t 5 10110
u 4 1010
v 5 11110
Prev 'm' table:
Table:
\\sp 2 00
, 4 1110
. 4 1100
a 3 011
c 5 11110
e 3 010
i 5 11111
m 5 10110
o 3 100
p 4 1101
s 5 10111
u 4 1010
Prev 'n' table:
Table:
\\sp 3 000
, 7 1111111
. 7 1111110
a 3 101
c 4 0110
d 3 100
e 3 110
g 5 11110
i 3 010
o 4 1110
s 5 01110
t 3 001
u 5 01111
v 6 111110
Prev 'o' table:
Table:
\\sp 3 010
, 6 111011
. 5 11100
b 6 111010
c 8 11111110
d 4 1101
i 5 11110
l 3 011
m 6 111110
n 2 00
q 7 1111110
r 2 10
s 4 1100
t 8 11111111
Prev 'p' table:
Table:
a 4 1100
e 2 01
h 6 111111
i 2 00
l 5 11110
o 3 101
r 4 1101
s 4 1110
t 6 111110
u 3 100
Prev 'q' table:
Table:
u 1 1
Prev 'r' table:
Table:
\\sp 3 000
, 6 011111
. 5 11110
a 3 010
b 6 101111
c 4 1101
d 5 10110
e 3 100
h 6 011110
i 3 001
m 6 111110
n 5 01110
o 4 1010
p 4 1100
q 7 1111110
r 7 1111111
s 6 101110
t 4 0110
u 4 1110
Prev 's' table:
Table:
\\sp 2 10
, 4 0110
. 4 1101
a 4 0011
c 5 01110
e 3 010
i 4 1100
l 6 011111
m 6 011110
o 6 111111
p 6 111110
q 4 0010
s 5 11110
t 3 000
u 4 1110
Prev 't' table:
Table:
\\sp 2 00
, 4 1100
. 4 1101
a 4 1010
e 3 100
i 3 010
o 4 1011
p 5 11110
r 4 1110
t 5 11111
u 3 011
Prev 'u' table:
Table:
\\sp 5 10111
, 8 11111110
. 9 111111111
a 5 11101
b 9 111111110
c 5 10110
d 7 1111110
e 3 010
g 6 111110
i 4 1010
l 3 110
m 3 011
n 5 11110
r 3 100
s 2 00
t 5 11100
Prev 'v' table:
Table:
a 3 110
e 1 0
i 2 10
o 4 1111
u 4 1110
Prev 'x' table:
Table:
\\sp 2 10
, 3 110
. 3 111
i 1 0
graph G {
	packmode="cluster";
/* Prev '\\n' tree: */
subgraph clusterG0 {
	label="Prev: \\n";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n0;
	n0 [label=""];
	n1;
	n1 [label="\\n"];
	n0 -- n1;
	n2;
	n2 [label=""];
	n3;
	n3 [label=""];
	n4;
	n4 [label=""];
	n5;
	n5 [label="P"];
	n4 -- n5;
	n6;
	n6 [label="N"];
	n4 -- n6;
	n3 -- n4;
	n7;
	n7 [label=""];
	n8;
	n8 [label=""];
	n9;
	n9 [label="A"];
	n8 -- n9;
	n10;
	n10 [label="M"];
	n8 -- n10;
	n7 -- n8;
	n11;
	n11 [label=""];
	n12;
	n12 [label="I"];
	n11 -- n12;
	n13;
	n13 [label=""];
	n14;
	n14 [label="U"];
	n13 -- n14;
	n15;
	n15 [label="Q"];
	n13 -- n15;
	n11 -- n13;
	n7 -- n11;
	n3 -- n7;
	n2 -- n3;
	n16;
	n16 [label=""];
	n17;
	n17 [label=""];
	n18;
	n18 [label="S"];
	n17 -- n18;
	n19;
	n19 [label="D"];
	n17 -- n19;
	n16 -- n17;
	n20;
	n20 [label=""];
	n21;
	n21 [label="C"];
	n20 -- n21;
	n22;
	n22 [label=""];
	n23;
	n23 [label="V"];
	n22 -- n23;
	n24;
	n24 [label=""];
	n25;
	n25 [label="E"];
	n24 -- n25;
	n26;
	n26 [label=""];
	n27;
	n27 [label="F"];
	n26 -- n27;
	n28;
	n28 [label="O"];
	n26 -- n28;
	n24 -- n26;
	n22 -- n24;
	n20 -- n22;
	n16 -- n20;
	n2 -- n16;
	n0 -- n2;
}
/* Prev '\\sp' tree: */
subgraph clusterG29 {
	label="Prev: \\sp";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n29;
	n29 [label=""];
	n30;
	n30 [label=""];
	n31;
	n31 [label=""];
	n32;
	n32 [label="e"];
	n31 -- n32;
	n33;
	n33 [label=""];
	n34;
	n34 [label="l"];
	n33 -- n34;
	n35;
	n35 [label="m"];
	n33 -- n35;
	n31 -- n33;
	n30 -- n31;
	n36;
	n36 [label=""];
	n37;
	n37 [label=""];
	n38;
	n38 [label="t"];
	n37 -- n38;
	n39;
	n39 [label="i"];
	n37 -- n39;
	n36 -- n37;
	n40;
	n40 [label=""];
	n41;
	n41 [label="p"];
	n40 -- n41;
	n42;
	n42 [label=""];
	n43;
	n43 [label=""];
	n44;
	n44 [label="D"];
	n43 -- n44;
	n45;
	n45 [label="S"];
	n43 -- n45;
	n42 -- n43;
	n46;
	n46 [label=""];
	n47;
	n47 [label="N"];
	n46 -- n47;
	n48;
	n48 [label=""];
	n49;
	n49 [label="A"];
	n48 -- n49;
	n50;
	n50 [label="j"];
	n48 -- n50;
	n46 -- n48;
	n42 -- n46;
	n40 -- n42;
	n36 -- n40;
	n30 -- n36;
	n29 -- n30;
	n51;
	n51 [label=""];
	n52;
	n52 [label=""];
	n53;
	n53 [label=""];
	n54;
	n54 [label="n"];
	n53 -- n54;
	n55;
	n55 [label="v"];
	n53 -- n55;
	n52 -- n53;
	n56;
	n56 [label=""];
	n57;
	n57 [label="s"];
	n56 -- n57;
	n58;
	n58 [label=""];
	n59;
	n59 [label=""];
	n60;
	n60 [label="P"];
	n59 -- n60;
	n61;
	n61 [label=""];
	n62;
	n62 [label="V"];
	n61 -- n62;
	n63;
	n63 [label="C"];
	n61 -- n63;
	n59 -- n61;
	n58 -- n59;
	n64;
	n64 [label=""];
	n65;
	n65 [label="r"];
	n64 -- n65;
	n66;
	n66 [label=""];
	n67;
	n67 [label="I"];
	n66 -- n67;
	n68;
	n68 [label=""];
	n69;
	n69 [label="E"];
	n68 -- n69;
	n70;
	n70 [label="F"];
	n68 -- n70;
	n66 -- n68;
	n64 -- n66;
	n58 -- n64;
	n56 -- n58;
	n52 -- n56;
	n51 -- n52;
	n71;
	n71 [label=""];
	n72;
	n72 [label=""];
	n73;
	n73 [label=""];
	n74;
	n74 [label="u"];
	n73 -- n74;
	n75;
	n75 [label="f"];
	n73 -- n75;
	n72 -- n73;
	n76;
	n76 [label=""];
	n77;
	n77 [label="c"];
	n76 -- n77;
	n78;
	n78 [label=""];
	n79;
	n79 [label="q"];
	n78 -- n79;
	n80;
	n80 [label="o"];
	n78 -- n80;
	n76 -- n78;
	n72 -- n76;
	n71 -- n72;
	n81;
	n81 [label=""];
	n82;
	n82 [label="a"];
	n81 -- n82;
	n83;
	n83 [label=""];
	n84;
	n84 [label="d"];
	n83 -- n84;
	n85;
	n85 [label=""];
	n86;
	n86 [label=""];
	n87;
	n87 [label="b"];
	n86 -- n87;
	n88;
	n88 [label=""];
	n89;
	n89 [label="Q"];
	n88 -- n89;
	n90;
	n90 [label="g"];
	n88 -- n90;
	n86 -- n88;
	n85 -- n86;
	n91;
	n91 [label=""];
	n92;
	n92 [label="M"];
	n91 -- n92;
	n93;
	n93 [label=""];
	n94;
	n94 [label="h"];
	n93 -- n94;
	n95;
	n95 [label=""];
	n96;
	n96 [label="U"];
	n95 -- n96;
	n97;
	n97 [label=""];
	n98;
	n98 [label="L"];
	n97 -- n98;
	n99;
	n99 [label="O"];
	n97 -- n99;
	n95 -- n97;
	n93 -- n95;
	n91 -- n93;
	n85 -- n91;
	n83 -- n85;
	n81 -- n83;
	n71 -- n81;
	n51 -- n71;
	n29 -- n51;
}
/* Prev ',' tree: */
subgraph clusterG100 {
	label="Prev: ,";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n100;
	n100 [label=""];
	n101;
	n101 [label="\\sp"];
	n100 -- n101;
	n102;
	n102 [label="\\sp"];
	n100 -- n102;
}
/* Prev '.' tree: */
subgraph clusterG103 {
	label="Prev: .";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n103;
	n103 [label=""];
	n104;
	n104 [label="\\n"];
	n103 -- n104;
	n105;
	n105 [label="\\sp"];
	n103 -- n105;
}
/* Prev ';' tree: */
subgraph clusterG106 {
	label="Prev: ;";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n106;
	n106 [label=""];
	n107;
	n107 [label="\\sp"];
	n106 -- n107;
	n108;
	n108 [label="\\sp"];
	n106 -- n108;
}
/* Prev 'A' tree: */
subgraph clusterG109 {
	label="Prev: A";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n109;
	n109 [label=""];
	n110;
	n110 [label="e"];
	n109 -- n110;
	n111;
	n111 [label="l"];
	n109 -- n111;
}
/* Prev 'C' tree: */
subgraph clusterG112 {
	label="Prev: C";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n112;
	n112 [label=""];
	n113;
	n113 [label="u"];
	n112 -- n113;
	n114;
	n114 [label=""];
	n115;
	n115 [label="l"];
	n114 -- n115;
	n116;
	n116 [label="r"];
	n114 -- n116;
	n112 -- n114;
}
/* Prev 'D' tree: */
subgraph clusterG117 {
	label="Prev: D";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n117;
	n117 [label=""];
	n118;
	n118 [label="u"];
	n117 -- n118;
	n119;
	n119 [label="o"];
	n117 -- n119;
}
/* Prev 'E' tree: */
subgraph clusterG120 {
	label="Prev: E";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n120;
	n120 [label=""];
	n121;
	n121 [label="t"];
	n120 -- n121;
	n122;
	n122 [label="t"];
	n120 -- n122;
}
/* Prev 'F' tree: */
subgraph clusterG123 {
	label="Prev: F";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n123;
	n123 [label=""];
	n124;
	n124 [label="u"];
	n123 -- n124;
	n125;
	n125 [label="u"];
	n123 -- n125;
}
/* Prev 'I' tree: */
subgraph clusterG126 {
	label="Prev: I";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n126;
	n126 [label=""];
	n127;
	n127 [label="n"];
	n126 -- n127;
	n128;
	n128 [label="n"];
	n126 -- n128;
}
/* Prev 'L' tree: */
subgraph clusterG129 {
	label="Prev: L";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n129;
	n129 [label=""];
	n130;
	n130 [label="o"];
	n129 -- n130;
	n131;
	n131 [label="o"];
	n129 -- n131;
}
/* Prev 'M' tree: */
subgraph clusterG132 {
	label="Prev: M";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n132;
	n132 [label=""];
	n133;
	n133 [label="o"];
	n132 -- n133;
	n134;
	n134 [label="a"];
	n132 -- n134;
}
/* Prev 'N' tree: */
subgraph clusterG135 {
	label="Prev: N";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n135;
	n135 [label=""];
	n136;
	n136 [label="a"];
	n135 -- n136;
	n137;
	n137 [label="u"];
	n135 -- n137;
}
/* Prev 'O' tree: */
subgraph clusterG138 {
	label="Prev: O";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n138;
	n138 [label=""];
	n139;
	n139 [label="r"];
	n138 -- n139;
	n140;
	n140 [label="r"];
	n138 -- n140;
}
/* Prev 'P' tree: */
subgraph clusterG141 {
	label="Prev: P";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n141;
	n141 [label=""];
	n142;
	n142 [label="r"];
	n141 -- n142;
	n143;
	n143 [label=""];
	n144;
	n144 [label="h"];
	n143 -- n144;
	n145;
	n145 [label="e"];
	n143 -- n145;
	n141 -- n143;
}
/* Prev 'Q' tree: */
subgraph clusterG146 {
	label="Prev: Q";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n146;
	n146 [label=""];
	n147;
	n147 [label="u"];
	n146 -- n147;
	n148;
	n148 [label="u"];
	n146 -- n148;
}
/* Prev 'S' tree: */
subgraph clusterG149 {
	label="Prev: S";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n149;
	n149 [label=""];
	n150;
	n150 [label="u"];
	n149 -- n150;
	n151;
	n151 [label="e"];
	n149 -- n151;
}
/* Prev 'U' tree: */
subgraph clusterG152 {
	label="Prev: U";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n152;
	n152 [label=""];
	n153;
	n153 [label="t"];
	n152 -- n153;
	n154;
	n154 [label="t"];
	n152 -- n154;
}
/* Prev 'V' tree: */
subgraph clusterG155 {
	label="Prev: V";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n155;
	n155 [label=""];
	n156;
	n156 [label="i"];
	n155 -- n156;
	n157;
	n157 [label="e"];
	n155 -- n157;
}
/* Prev 'a' tree: */
subgraph clusterG158 {
	label="Prev: a";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n158;
	n158 [label=""];
	n159;
	n159 [label=""];
	n160;
	n160 [label=""];
	n161;
	n161 [label="c"];
	n160 -- n161;
	n162;
	n162 [label=""];
	n163;
	n163 [label="e"];
	n162 -- n163;
	n164;
	n164 [label="l"];
	n162 -- n164;
	n160 -- n162;
	n159 -- n160;
	n165;
	n165 [label=""];
	n166;
	n166 [label="t"];
	n165 -- n166;
	n167;
	n167 [label=""];
	n168;
	n168 [label="s"];
	n167 -- n168;
	n169;
	n169 [label=""];
	n170;
	n170 [label="g"];
	n169 -- n170;
	n171;
	n171 [label="."];
	n169 -- n171;
	n167 -- n169;
	n165 -- n167;
	n159 -- n165;
	n158 -- n159;
	n172;
	n172 [label=""];
	n173;
	n173 [label=""];
	n174;
	n174 [label="m"];
	n173 -- n174;
	n175;
	n175 [label=""];
	n176;
	n176 [label="r"];
	n175 -- n176;
	n177;
	n177 [label="u"];
	n175 -- n177;
	n173 -- n175;
	n172 -- n173;
	n178;
	n178 [label=""];
	n179;
	n179 [label="\\sp"];
	n178 -- n179;
	n180;
	n180 [label=""];
	n181;
	n181 [label=""];
	n182;
	n182 [label="p"];
	n181 -- n182;
	n183;
	n183 [label=""];
	n184;
	n184 [label="d"];
	n183 -- n184;
	n185;
	n185 [label=""];
	n186;
	n186 [label="x"];
	n185 -- n186;
	n187;
	n187 [label="v"];
	n185 -- n187;
	n183 -- n185;
	n181 -- n183;
	n180 -- n181;
	n188;
	n188 [label=""];
	n189;
	n189 [label="n"];
	n188 -- n189;
	n190;
	n190 [label=""];
	n191;
	n191 [label=","];
	n190 -- n191;
	n192;
	n192 [label=""];
	n193;
	n193 [label="o"];
	n192 -- n193;
	n194;
	n194 [label="b"];
	n192 -- n194;
	n190 -- n192;
	n188 -- n190;
	n180 -- n188;
	n178 -- n180;
	n172 -- n178;
	n158 -- n172;
}
/* Prev 'b' tree: */
subgraph clusterG195 {
	label="Prev: b";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n195;
	n195 [label=""];
	n196;
	n196 [label="u"];
	n195 -- n196;
	n197;
	n197 [label=""];
	n198;
	n198 [label="i"];
	n197 -- n198;
	n199;
	n199 [label=""];
	n200;
	n200 [label="e"];
	n199 -- n200;
	n201;
	n201 [label=""];
	n202;
	n202 [label="h"];
	n201 -- n202;
	n203;
	n203 [label=""];
	n204;
	n204 [label="o"];
	n203 -- n204;
	n205;
	n205 [label="l"];
	n203 -- n205;
	n201 -- n203;
	n199 -- n201;
	n197 -- n199;
	n195 -- n197;
}
/* Prev 'c' tree: */
subgraph clusterG206 {
	label="Prev: c";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n206;
	n206 [label=""];
	n207;
	n207 [label=""];
	n208;
	n208 [label="\\sp"];
	n207 -- n208;
	n209;
	n209 [label="i"];
	n207 -- n209;
	n206 -- n207;
	n210;
	n210 [label=""];
	n211;
	n211 [label=""];
	n212;
	n212 [label="t"];
	n211 -- n212;
	n213;
	n213 [label="o"];
	n211 -- n213;
	n210 -- n211;
	n214;
	n214 [label=""];
	n215;
	n215 [label="u"];
	n214 -- n215;
	n216;
	n216 [label=""];
	n217;
	n217 [label="e"];
	n216 -- n217;
	n218;
	n218 [label=""];
	n219;
	n219 [label="c"];
	n218 -- n219;
	n220;
	n220 [label=""];
	n221;
	n221 [label=","];
	n220 -- n221;
	n222;
	n222 [label="."];
	n220 -- n222;
	n218 -- n220;
	n216 -- n218;
	n214 -- n216;
	n210 -- n214;
	n206 -- n210;
}
/* Prev 'd' tree: */
subgraph clusterG223 {
	label="Prev: d";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n223;
	n223 [label=""];
	n224;
	n224 [label=""];
	n225;
	n225 [label="u"];
	n224 -- n225;
	n226;
	n226 [label="\\sp"];
	n224 -- n226;
	n223 -- n224;
	n227;
	n227 [label=""];
	n228;
	n228 [label="i"];
	n227 -- n228;
	n229;
	n229 [label=""];
	n230;
	n230 [label="a"];
	n229 -- n230;
	n231;
	n231 [label=""];
	n232;
	n232 [label="o"];
	n231 -- n232;
	n233;
	n233 [label=""];
	n234;
	n234 [label="r"];
	n233 -- n234;
	n235;
	n235 [label=""];
	n236;
	n236 [label="."];
	n235 -- n236;
	n237;
	n237 [label=","];
	n235 -- n237;
	n233 -- n235;
	n231 -- n233;
	n229 -- n231;
	n227 -- n229;
	n223 -- n227;
}
/* Prev 'e' tree: */
subgraph clusterG238 {
	label="Prev: e";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n238;
	n238 [label=""];
	n239;
	n239 [label=""];
	n240;
	n240 [label=""];
	n241;
	n241 [label="r"];
	n240 -- n241;
	n242;
	n242 [label="\\sp"];
	n240 -- n242;
	n239 -- n240;
	n243;
	n243 [label=""];
	n244;
	n244 [label="l"];
	n243 -- n244;
	n245;
	n245 [label=""];
	n246;
	n246 [label="m"];
	n245 -- n246;
	n247;
	n247 [label="c"];
	n245 -- n247;
	n243 -- n245;
	n239 -- n243;
	n238 -- n239;
	n248;
	n248 [label=""];
	n249;
	n249 [label=""];
	n250;
	n250 [label="t"];
	n249 -- n250;
	n251;
	n251 [label="n"];
	n249 -- n251;
	n248 -- n249;
	n252;
	n252 [label=""];
	n253;
	n253 [label=""];
	n254;
	n254 [label=""];
	n255;
	n255 [label="g"];
	n254 -- n255;
	n256;
	n256 [label="u"];
	n254 -- n256;
	n253 -- n254;
	n257;
	n257 [label=""];
	n258;
	n258 [label="d"];
	n257 -- n258;
	n259;
	n259 [label=""];
	n260;
	n260 [label=""];
	n261;
	n261 [label="i"];
	n260 -- n261;
	n262;
	n262 [label="h"];
	n260 -- n262;
	n259 -- n260;
	n263;
	n263 [label=""];
	n264;
	n264 [label="e"];
	n263 -- n264;
	n265;
	n265 [label="f"];
	n263 -- n265;
	n259 -- n263;
	n257 -- n259;
	n253 -- n257;
	n252 -- n253;
	n266;
	n266 [label=""];
	n267;
	n267 [label="s"];
	n266 -- n267;
	n268;
	n268 [label=""];
	n269;
	n269 [label=""];
	n270;
	n270 [label="."];
	n269 -- n270;
	n271;
	n271 [label="q"];
	n269 -- n271;
	n268 -- n269;
	n272;
	n272 [label=""];
	n273;
	n273 [label=""];
	n274;
	n274 [label=","];
	n273 -- n274;
	n275;
	n275 [label="o"];
	n273 -- n275;
	n272 -- n273;
	n276;
	n276 [label=""];
	n277;
	n277 [label="x"];
	n276 -- n277;
	n278;
	n278 [label=""];
	n279;
	n279 [label="a"];
	n278 -- n279;
	n280;
	n280 [label=""];
	n281;
	n281 [label="p"];
	n280 -- n281;
	n282;
	n282 [label=";"];
	n280 -- n282;
	n278 -- n280;
	n276 -- n278;
	n272 -- n276;
	n268 -- n272;
	n266 -- n268;
	n252 -- n266;
	n248 -- n252;
	n238 -- n248;
}
/* Prev 'f' tree: */
subgraph clusterG283 {
	label="Prev: f";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n283;
	n283 [label=""];
	n284;
	n284 [label=""];
	n285;
	n285 [label="a"];
	n284 -- n285;
	n286;
	n286 [label="e"];
	n284 -- n286;
	n283 -- n284;
	n287;
	n287 [label=""];
	n288;
	n288 [label="i"];
	n287 -- n288;
	n289;
	n289 [label=""];
	n290;
	n290 [label="r"];
	n289 -- n290;
	n291;
	n291 [label="f"];
	n289 -- n291;
	n287 -- n289;
	n283 -- n287;
}
/* Prev 'g' tree: */
subgraph clusterG292 {
	label="Prev: g";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n292;
	n292 [label=""];
	n293;
	n293 [label=""];
	n294;
	n294 [label="u"];
	n293 -- n294;
	n295;
	n295 [label="e"];
	n293 -- n295;
	n292 -- n293;
	n296;
	n296 [label=""];
	n297;
	n297 [label="i"];
	n296 -- n297;
	n298;
	n298 [label=""];
	n299;
	n299 [label="n"];
	n298 -- n299;
	n300;
	n300 [label=""];
	n301;
	n301 [label="\\sp"];
	n300 -- n301;
	n302;
	n302 [label="r"];
	n300 -- n302;
	n298 -- n300;
	n296 -- n298;
	n292 -- n296;
}
/* Prev 'h' tree: */
subgraph clusterG303 {
	label="Prev: h";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n303;
	n303 [label=""];
	n304;
	n304 [label=""];
	n305;
	n305 [label="a"];
	n304 -- n305;
	n306;
	n306 [label=""];
	n307;
	n307 [label="e"];
	n306 -- n307;
	n308;
	n308 [label="o"];
	n306 -- n308;
	n304 -- n306;
	n303 -- n304;
	n309;
	n309 [label=""];
	n310;
	n310 [label="i"];
	n309 -- n310;
	n311;
	n311 [label=""];
	n312;
	n312 [label="\\sp"];
	n311 -- n312;
	n313;
	n313 [label=""];
	n314;
	n314 [label=","];
	n313 -- n314;
	n315;
	n315 [label="."];
	n313 -- n315;
	n311 -- n313;
	n309 -- n311;
	n303 -- n309;
}
/* Prev 'i' tree: */
subgraph clusterG316 {
	label="Prev: i";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n316;
	n316 [label=""];
	n317;
	n317 [label=""];
	n318;
	n318 [label="s"];
	n317 -- n318;
	n319;
	n319 [label=""];
	n320;
	n320 [label="n"];
	n319 -- n320;
	n321;
	n321 [label=""];
	n322;
	n322 [label="d"];
	n321 -- n322;
	n323;
	n323 [label="m"];
	n321 -- n323;
	n319 -- n321;
	n317 -- n319;
	n316 -- n317;
	n324;
	n324 [label=""];
	n325;
	n325 [label=""];
	n326;
	n326 [label="t"];
	n325 -- n326;
	n327;
	n327 [label=""];
	n328;
	n328 [label="c"];
	n327 -- n328;
	n329;
	n329 [label=""];
	n330;
	n330 [label="p"];
	n329 -- n330;
	n331;
	n331 [label=""];
	n332;
	n332 [label="v"];
	n331 -- n332;
	n333;
	n333 [label=""];
	n334;
	n334 [label=","];
	n333 -- n334;
	n335;
	n335 [label="f"];
	n333 -- n335;
	n331 -- n333;
	n329 -- n331;
	n327 -- n329;
	n325 -- n327;
	n324 -- n325;
	n336;
	n336 [label=""];
	n337;
	n337 [label=""];
	n338;
	n338 [label=""];
	n339;
	n339 [label="q"];
	n338 -- n339;
	n340;
	n340 [label="e"];
	n338 -- n340;
	n337 -- n338;
	n341;
	n341 [label=""];
	n342;
	n342 [label="\\sp"];
	n341 -- n342;
	n343;
	n343 [label=""];
	n344;
	n344 [label="u"];
	n343 -- n344;
	n345;
	n345 [label="l"];
	n343 -- n345;
	n341 -- n343;
	n337 -- n341;
	n336 -- n337;
	n346;
	n346 [label=""];
	n347;
	n347 [label="b"];
	n346 -- n347;
	n348;
	n348 [label=""];
	n349;
	n349 [label="a"];
	n348 -- n349;
	n350;
	n350 [label=""];
	n351;
	n351 [label="g"];
	n350 -- n351;
	n352;
	n352 [label=""];
	n353;
	n353 [label="."];
	n352 -- n353;
	n354;
	n354 [label="o"];
	n352 -- n354;
	n350 -- n352;
	n348 -- n350;
	n346 -- n348;
	n336 -- n346;
	n324 -- n336;
	n316 -- n324;
}
/* Prev 'j' tree: */
subgraph clusterG355 {
	label="Prev: j";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n355;
	n355 [label=""];
	n356;
	n356 [label="u"];
	n355 -- n356;
	n357;
	n357 [label="u"];
	n355 -- n357;
}
/* Prev 'l' tree: */
subgraph clusterG358 {
	label="Prev: l";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n358;
	n358 [label=""];
	n359;
	n359 [label=""];
	n360;
	n360 [label="a"];
	n359 -- n360;
	n361;
	n361 [label="i"];
	n359 -- n361;
	n358 -- n359;
	n362;
	n362 [label=""];
	n363;
	n363 [label=""];
	n364;
	n364 [label="l"];
	n363 -- n364;
	n365;
	n365 [label=""];
	n366;
	n366 [label="u"];
	n365 -- n366;
	n367;
	n367 [label=""];
	n368;
	n368 [label="t"];
	n367 -- n368;
	n369;
	n369 [label="\\sp"];
	n367 -- n369;
	n365 -- n367;
	n363 -- n365;
	n362 -- n363;
	n370;
	n370 [label=""];
	n371;
	n371 [label="e"];
	n370 -- n371;
	n372;
	n372 [label=""];
	n373;
	n373 [label="o"];
	n372 -- n373;
	n374;
	n374 [label=""];
	n375;
	n375 [label="v"];
	n374 -- n375;
	n376;
	n376 [label=""];
	n377;
	n377 [label="p"];
	n376 -- n377;
	n378;
	n378 [label=""];
	n379;
	n379 [label="."];
	n378 -- n379;
	n380;
	n380 [label=","];
	n378 -- n380;
	n376 -- n378;
	n374 -- n376;
	n372 -- n374;
	n370 -- n372;
	n362 -- n370;
	n358 -- n362;
}
/* Prev 'm' tree: */
subgraph clusterG381 {
	label="Prev: m";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n381;
	n381 [label=""];
	n382;
	n382 [label=""];
	n383;
	n383 [label="\\sp"];
	n382 -- n383;
	n384;
	n384 [label=""];
	n385;
	n385 [label="e"];
	n384 -- n385;
	n386;
	n386 [label="a"];
	n384 -- n386;
	n382 -- n384;
	n381 -- n382;
	n387;
	n387 [label=""];
	n388;
	n388 [label=""];
	n389;
	n389 [label="o"];
	n388 -- n389;
	n390;
	n390 [label=""];
	n391;
	n391 [label="u"];
	n390 -- n391;
	n392;
	n392 [label=""];
	n393;
	n393 [label="m"];
	n392 -- n393;
	n394;
	n394 [label="s"];
	n392 -- n394;
	n390 -- n392;
	n388 -- n390;
	n387 -- n388;
	n395;
	n395 [label=""];
	n396;
	n396 [label=""];
	n397;
	n397 [label="."];
	n396 -- n397;
	n398;
	n398 [label="p"];
	n396 -- n398;
	n395 -- n396;
	n399;
	n399 [label=""];
	n400;
	n400 [label=","];
	n399 -- n400;
	n401;
	n401 [label=""];
	n402;
	n402 [label="c"];
	n401 -- n402;
	n403;
	n403 [label="i"];
	n401 -- n403;
	n399 -- n401;
	n395 -- n399;
	n387 -- n395;
	n381 -- n387;
}
/* Prev 'n' tree: */
subgraph clusterG404 {
	label="Prev: n";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n404;
	n404 [label=""];
	n405;
	n405 [label=""];
	n406;
	n406 [label=""];
	n407;
	n407 [label="\\sp"];
	n406 -- n407;
	n408;
	n408 [label="t"];
	n406 -- n408;
	n405 -- n406;
	n409;
	n409 [label=""];
	n410;
	n410 [label="i"];
	n409 -- n410;
	n411;
	n411 [label=""];
	n412;
	n412 [label="c"];
	n411 -- n412;
	n413;
	n413 [label=""];
	n414;
	n414 [label="s"];
	n413 -- n414;
	n415;
	n415 [label="u"];
	n413 -- n415;
	n411 -- n413;
	n409 -- n411;
	n405 -- n409;
	n404 -- n405;
	n416;
	n416 [label=""];
	n417;
	n417 [label=""];
	n418;
	n418 [label="d"];
	n417 -- n418;
	n419;
	n419 [label="a"];
	n417 -- n419;
	n416 -- n417;
	n420;
	n420 [label=""];
	n421;
	n421 [label="e"];
	n420 -- n421;
	n422;
	n422 [label=""];
	n423;
	n423 [label="o"];
	n422 -- n423;
	n424;
	n424 [label=""];
	n425;
	n425 [label="g"];
	n424 -- n425;
	n426;
	n426 [label=""];
	n427;
	n427 [label="v"];
	n426 -- n427;
	n428;
	n428 [label=""];
	n429;
	n429 [label="."];
	n428 -- n429;
	n430;
	n430 [label=","];
	n428 -- n430;
	n426 -- n428;
	n424 -- n426;
	n422 -- n424;
	n420 -- n422;
	n416 -- n420;
	n404 -- n416;
}
/* Prev 'o' tree: */
subgraph clusterG431 {
	label="Prev: o";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n431;
	n431 [label=""];
	n432;
	n432 [label=""];
	n433;
	n433 [label="n"];
	n432 -- n433;
	n434;
	n434 [label=""];
	n435;
	n435 [label="\\sp"];
	n434 -- n435;
	n436;
	n436 [label="l"];
	n434 -- n436;
	n432 -- n434;
	n431 -- n432;
	n437;
	n437 [label=""];
	n438;
	n438 [label="r"];
	n437 -- n438;
	n439;
	n439 [label=""];
	n440;
	n440 [label=""];
	n441;
	n441 [label="s"];
	n440 -- n441;
	n442;
	n442 [label="d"];
	n440 -- n442;
	n439 -- n440;
	n443;
	n443 [label=""];
	n444;
	n444 [label=""];
	n445;
	n445 [label="."];
	n444 -- n445;
	n446;
	n446 [label=""];
	n447;
	n447 [label="b"];
	n446 -- n447;
	n448;
	n448 [label=","];
	n446 -- n448;
	n444 -- n446;
	n443 -- n444;
	n449;
	n449 [label=""];
	n450;
	n450 [label="i"];
	n449 -- n450;
	n451;
	n451 [label=""];
	n452;
	n452 [label="m"];
	n451 -- n452;
	n453;
	n453 [label=""];
	n454;
	n454 [label="q"];
	n453 -- n454;
	n455;
	n455 [label=""];
	n456;
	n456 [label="c"];
	n455 -- n456;
	n457;
	n457 [label="t"];
	n455 -- n457;
	n453 -- n455;
	n451 -- n453;
	n449 -- n451;
	n443 -- n449;
	n439 -- n443;
	n437 -- n439;
	n431 -- n437;
}
/* Prev 'p' tree: */
subgraph clusterG458 {
	label="Prev: p";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n458;
	n458 [label=""];
	n459;
	n459 [label=""];
	n460;
	n460 [label="i"];
	n459 -- n460;
	n461;
	n461 [label="e"];
	n459 -- n461;
	n458 -- n459;
	n462;
	n462 [label=""];
	n463;
	n463 [label=""];
	n464;
	n464 [label="u"];
	n463 -- n464;
	n465;
	n465 [label="o"];
	n463 -- n465;
	n462 -- n463;
	n466;
	n466 [label=""];
	n467;
	n467 [label=""];
	n468;
	n468 [label="a"];
	n467 -- n468;
	n469;
	n469 [label="r"];
	n467 -- n469;
	n466 -- n467;
	n470;
	n470 [label=""];
	n471;
	n471 [label="s"];
	n470 -- n471;
	n472;
	n472 [label=""];
	n473;
	n473 [label="l"];
	n472 -- n473;
	n474;
	n474 [label=""];
	n475;
	n475 [label="t"];
	n474 -- n475;
	n476;
	n476 [label="h"];
	n474 -- n476;
	n472 -- n474;
	n470 -- n472;
	n466 -- n470;
	n462 -- n466;
	n458 -- n462;
}
/* Prev 'q' tree: */
subgraph clusterG477 {
	label="Prev: q";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n477;
	n477 [label=""];
	n478;
	n478 [label="u"];
	n477 -- n478;
	n479;
	n479 [label="u"];
	n477 -- n479;
}
/* Prev 'r' tree: */
subgraph clusterG480 {
	label="Prev: r";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n480;
	n480 [label=""];
	n481;
	n481 [label=""];
	n482;
	n482 [label=""];
	n483;
	n483 [label="\\sp"];
	n482 -- n483;
	n484;
	n484 [label="i"];
	n482 -- n484;
	n481 -- n482;
	n485;
	n485 [label=""];
	n486;
	n486 [label="a"];
	n485 -- n486;
	n487;
	n487 [label=""];
	n488;
	n488 [label="t"];
	n487 -- n488;
	n489;
	n489 [label=""];
	n490;
	n490 [label="n"];
	n489 -- n490;
	n491;
	n491 [label=""];
	n492;
	n492 [label="h"];
	n491 -- n492;
	n493;
	n493 [label=","];
	n491 -- n493;
	n489 -- n491;
	n487 -- n489;
	n485 -- n487;
	n481 -- n485;
	n480 -- n481;
	n494;
	n494 [label=""];
	n495;
	n495 [label=""];
	n496;
	n496 [label="e"];
	n495 -- n496;
	n497;
	n497 [label=""];
	n498;
	n498 [label="o"];
	n497 -- n498;
	n499;
	n499 [label=""];
	n500;
	n500 [label="d"];
	n499 -- n500;
	n501;
	n501 [label=""];
	n502;
	n502 [label="s"];
	n501 -- n502;
	n503;
	n503 [label="b"];
	n501 -- n503;
	n499 -- n501;
	n497 -- n499;
	n495 -- n497;
	n494 -- n495;
	n504;
	n504 [label=""];
	n505;
	n505 [label=""];
	n506;
	n506 [label="p"];
	n505 -- n506;
	n507;
	n507 [label="c"];
	n505 -- n507;
	n504 -- n505;
	n508;
	n508 [label=""];
	n509;
	n509 [label="u"];
	n508 -- n509;
	n510;
	n510 [label=""];
	n511;
	n511 [label="."];
	n510 -- n511;
	n512;
	n512 [label=""];
	n513;
	n513 [label="m"];
	n512 -- n513;
	n514;
	n514 [label=""];
	n515;
	n515 [label="q"];
	n514 -- n515;
	n516;
	n516 [label="r"];
	n514 -- n516;
	n512 -- n514;
	n510 -- n512;
	n508 -- n510;
	n504 -- n508;
	n494 -- n504;
	n480 -- n494;
}
/* Prev 's' tree: */
subgraph clusterG517 {
	label="Prev: s";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n517;
	n517 [label=""];
	n518;
	n518 [label=""];
	n519;
	n519 [label=""];
	n520;
	n520 [label="t"];
	n519 -- n520;
	n521;
	n521 [label=""];
	n522;
	n522 [label="q"];
	n521 -- n522;
	n523;
	n523 [label="a"];
	n521 -- n523;
	n519 -- n521;
	n518 -- n519;
	n524;
	n524 [label=""];
	n525;
	n525 [label="e"];
	n524 -- n525;
	n526;
	n526 [label=""];
	n527;
	n527 [label=","];
	n526 -- n527;
	n528;
	n528 [label=""];
	n529;
	n529 [label="c"];
	n528 -- n529;
	n530;
	n530 [label=""];
	n531;
	n531 [label="m"];
	n530 -- n531;
	n532;
	n532 [label="l"];
	n530 -- n532;
	n528 -- n530;
	n526 -- n528;
	n524 -- n526;
	n518 -- n524;
	n517 -- n518;
	n533;
	n533 [label=""];
	n534;
	n534 [label="\\sp"];
	n533 -- n534;
	n535;
	n535 [label=""];
	n536;
	n536 [label=""];
	n537;
	n537 [label="i"];
	n536 -- n537;
	n538;
	n538 [label="."];
	n536 -- n538;
	n535 -- n536;
	n539;
	n539 [label=""];
	n540;
	n540 [label="u"];
	n539 -- n540;
	n541;
	n541 [label=""];
	n542;
	n542 [label="s"];
	n541 -- n542;
	n543;
	n543 [label=""];
	n544;
	n544 [label="p"];
	n543 -- n544;
	n545;
	n545 [label="o"];
	n543 -- n545;
	n541 -- n543;
	n539 -- n541;
	n535 -- n539;
	n533 -- n535;
	n517 -- n533;
}
/* Prev 't' tree: */
subgraph clusterG546 {
	label="Prev: t";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n546;
	n546 [label=""];
	n547;
	n547 [label=""];
	n548;
	n548 [label="\\sp"];
	n547 -- n548;
	n549;
	n549 [label=""];
	n550;
	n550 [label="i"];
	n549 -- n550;
	n551;
	n551 [label="u"];
	n549 -- n551;
	n547 -- n549;
	n546 -- n547;
	n552;
	n552 [label=""];
	n553;
	n553 [label=""];
	n554;
	n554 [label="e"];
	n553 -- n554;
	n555;
	n555 [label=""];
	n556;
	n556 [label="a"];
	n555 -- n556;
	n557;
	n557 [label="o"];
	n555 -- n557;
	n553 -- n555;
	n552 -- n553;
	n558;
	n558 [label=""];
	n559;
	n559 [label=""];
	n560;
	n560 [label=","];
	n559 -- n560;
	n561;
	n561 [label="."];
	n559 -- n561;
	n558 -- n559;
	n562;
	n562 [label=""];
	n563;
	n563 [label="r"];
	n562 -- n563;
	n564;
	n564 [label=""];
	n565;
	n565 [label="p"];
	n564 -- n565;
	n566;
	n566 [label="t"];
	n564 -- n566;
	n562 -- n564;
	n558 -- n562;
	n552 -- n558;
	n546 -- n552;
}
/* Prev 'u' tree: */
subgraph clusterG567 {
	label="Prev: u";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n567;
	n567 [label=""];
	n568;
	n568 [label=""];
	n569;
	n569 [label="s"];
	n568 -- n569;
	n570;
	n570 [label=""];
	n571;
	n571 [label="e"];
	n570 -- n571;
	n572;
	n572 [label="m"];
	n570 -- n572;
	n568 -- n570;
	n567 -- n568;
	n573;
	n573 [label=""];
	n574;
	n574 [label=""];
	n575;
	n575 [label="r"];
	n574 -- n575;
	n576;
	n576 [label=""];
	n577;
	n577 [label="i"];
	n576 -- n577;
	n578;
	n578 [label=""];
	n579;
	n579 [label="c"];
	n578 -- n579;
	n580;
	n580 [label="\\sp"];
	n578 -- n580;
	n576 -- n578;
	n574 -- n576;
	n573 -- n574;
	n581;
	n581 [label=""];
	n582;
	n582 [label="l"];
	n581 -- n582;
	n583;
	n583 [label=""];
	n584;
	n584 [label=""];
	n585;
	n585 [label="t"];
	n584 -- n585;
	n586;
	n586 [label="a"];
	n584 -- n586;
	n583 -- n584;
	n587;
	n587 [label=""];
	n588;
	n588 [label="n"];
	n587 -- n588;
	n589;
	n589 [label=""];
	n590;
	n590 [label="g"];
	n589 -- n590;
	n591;
	n591 [label=""];
	n592;
	n592 [label="d"];
	n591 -- n592;
	n593;
	n593 [label=""];
	n594;
	n594 [label=","];
	n593 -- n594;
	n595;
	n595 [label=""];
	n596;
	n596 [label="b"];
	n595 -- n596;
	n597;
	n597 [label="."];
	n595 -- n597;
	n593 -- n595;
	n591 -- n593;
	n589 -- n591;
	n587 -- n589;
	n583 -- n587;
	n581 -- n583;
	n573 -- n581;
	n567 -- n573;
}
/* Prev 'v' tree: */
subgraph clusterG598 {
	label="Prev: v";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n598;
	n598 [label=""];
	n599;
	n599 [label="e"];
	n598 -- n599;
	n600;
	n600 [label=""];
	n601;
	n601 [label="i"];
	n600 -- n601;
	n602;
	n602 [label=""];
	n603;
	n603 [label="a"];
	n602 -- n603;
	n604;
	n604 [label=""];
	n605;
	n605 [label="u"];
	n604 -- n605;
	n606;
	n606 [label="o"];
	n604 -- n606;
	n602 -- n604;
	n600 -- n602;
	n598 -- n600;
}
/* Prev 'x' tree: */
subgraph clusterG607 {
	label="Prev: x";
	color=invis;
	nodesep=0.3;
	ranksep=0.2;
	node [shape=circle, fixedsize=true];
	edge [arrowsize=0.8];
	n607;
	n607 [label=""];
	n608;
	n608 [label="i"];
	n607 -- n608;
	n609;
	n609 [label=""];
	n610;
	n610 [label="\\sp"];
	n609 -- n610;
	n611;
	n611 [label=""];
	n612;
	n612 [label=","];
	n611 -- n612;
	n613;
	n613 [label="."];
	n611 -- n613;
	n609 -- n611;
	n607 -- n609;
}
}
