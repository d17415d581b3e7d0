]Vj\%Ŋθ
rхذibShAYW%v*
%
^UOS'`