"""Configuration texts for the config-parser parity tests (tests/test_config.py) and the generator of
tests/golden/config_parse.json.  Written for this repo: they follow the layout the reference's shell scripts
emit for each tool (a gas list, one section per gas, multi-line quoted file lists, numeric vectors) and then
walk the corners of the grammar one by one."""

CASES = {
    "find_g_points_lw": '''# generated for the longwave g-point search
append_path "/data/mmm/lw:/work/lw_spectra:/work/lw_order"
iprofile 0
averaging_method "transmission"
tolerance_tolerance 0.01
flux_weight 0.0
min_pressure 2.0
max_iterations 60
heating_rate_tolerance 0.047 0.031

gases h2o o3 co2

\\begin h2o
  # median water vapour
  input spectra_h2o_median.nc
  reordering_input order_narrow_h2o.nc
  background_input "spectra_composite_minimum.nc
spectra_o3_minimum.nc"
  min_g_points 1 2 1
\\end h2o

\\begin o3
  input spectra_o3_median.nc
  reordering_input order_narrow_o3.nc
  background_input "spectra_composite_minimum.nc
spectra_h2o_minimum.nc"
\\end o3

\\begin co2
  input spectra_co2_present.nc
  reordering_input order_narrow_co2.nc
  background_input "spectra_h2o_minimum.nc
spectra_o3_minimum.nc
spectra_ch4_present.nc
spectra_n2o_present.nc"
  background_conc -1 -1 350e-9 190e-9
  min_scaling 0.18   # collapses to 0.5 in the tool
  max_scaling 1.8
\\end co2
''',
    "create_lut": '''prepend_path /work/ckd
input gpoints_fsck.nc
output raw_definition.nc
gases composite h2o o3
temperature_planck 120 121 122.5 123
averaging_method "transmission-2"
save_min_max yes
\\begin composite
  conc_dependence none
  input "well_mixed_present.nc"
\\end composite
\\begin h2o
  conc_dependence lut
  input "h2o_a.nc  h2o_b.nc
   h2o_c.nc"
  iprofile 0 1 2
\\end h2o
\\begin o3
  conc_dependence linear
  input o3_median.nc
\\end o3
''',
    "optimize_lut": '''input raw_definition.nc
output ckd_definition.nc
training_input "fluxes_present.nc fluxes_4xco2.nc"
gases h2o o3
flux_weight 0.2
flux_profile_weight 0.0
broadband_weight .8
prior_error 8.0
temperature_corr 0.8
pressure_corr 0.8
conc_corr 0.8
max_iterations 3000
convergence_criterion 0.02
negative_optical_depth_penalty 1e4
relative_to "fluxes_rel_415.nc"
bounded FALSE
debug_partition No
use_prior 0
''',
    "quotes_and_braces": '''single 'a b  c'   trailing_is_a_new_param
double "x # not a comment
y"
curly { 1 2 # stripped comment
  3 4 }
empty ""
bare
bare_comment # nothing
name#glued comment
spaces    value with   inner   spaces     \t
tab\tseparated\tvalue
''',
    "dims_refs_tables": '''v[3] 1 2 3
mat[2][3] {1 2 3
 4 5 6}
base 42
copy $base
missing $nothing
(a b[2] c[2][2]) {1 2 3
 4 5 6
 7}
(lonely other) 9
BASE 43
v[5] 9 8
''',
    "sections_nested": '''top 1
\\begin outer
  x 1
  \\begin inner
    x 2
    y "p q"
  \\end inner
  x 3
  z[2] 5 6
\\end
\\unknown ignored
after 7
\\begin Outer
  w 8
\\end OUTER
''',
    "numbers": '''i1 12abc
i2 abc
i3 -7 8 9
r1 1.5e3x
r2 .5
r3 1e400
r4 nan
r5 0x10
b1 false
b2 No
b3 0.0
b4 nothing
b5 FALSEHOOD
b6 00
vec 1 2 three 4
ivec 1 2.5 3
lead   {  7 8 }
''',
    "crlf": "a 1\r\nb two words\r\n\\begin s\r\n c 3\r\n\\end s\r\n",
    "no_final_newline": "a 1\nb",
    "only_comments": "# nothing here\n\n   # really\n",
}

# argv vectors (argv[0] first); "@dir" is replaced by the directory holding the case files
ARGV_CASES = {
    "plain": ["exe", "iprofile=3", "@dir/find_g_points_lw.cfg", "output=out.nc"],
    "override_and_flag": ["exe", "-verbose", "@dir/optimize_lut.cfg", "flux_weight=0.5", "h2o.input=cmd.nc", "a=b=c"],
    "no_cfg": ["exe", "input=in.nc", "data.nc", "other.txt", "x=1"],
    "hyphen_quirk": ["exe", "--", "-strange.cfg", "@dir/numbers.cfg"],
    "reference_arg": ["exe", "@dir/dims_refs_tables.cfg", "q=$base", "second.cfg"],
    "single_hyphen": ["exe", "-", "k=v"],
}

# \include: files written next to each other; the entry point is main.cfg
INCLUDE_FILES = {
    "main.cfg": "first 1\n\\begin sec\n\\include sub/inc.cfg\n\\end sec\nlast 4\n",
    "sub/inc.cfg": "second 2\n\\include deeper.cfg\n",
    "sub/deeper.cfg": "third 3\n",
}

# texts the parser must reject
ERROR_CASES = {
    "end_without_begin": "a 1\n\\end\n",
    "end_mismatch": "\\begin a\nx 1\n\\end b\n",
    "unterminated_section": "\\begin a\nx 1\n",
    "include_nothing": "\\include\n",
    "include_missing": "\\include no_such_file.cfg\n",
}
