"""Physical constants, verbatim from the reference (src/constants.jl:1-26).

The Boltzmann constant is the CODATA-2014 value while h is the 2019 SI value -- kept on purpose.
"""
c = 299792458.0          # speed of light [m/s]
h = 6.62607015e-34       # Planck constant [J s]
k = 1.38064852e-23       # Boltzmann constant [J/K]
sigma_sb = 5.67037442e-8 # Stefan-Boltzmann constant [W/m^2/K^4]
R = 8.31446262           # gas constant [J/K/mole]
atm = 101325.0           # Pa in 1 atm
Na = 6.02214076e23       # Avogadro's number
Da = 1.66053907e-27      # Dalton [kg]
G = 6.6743e-11           # gravitational constant
Lo2 = 7.21879268e38      # Loschmidt number squared [molecules^2/cm^6]
Tref = 296.0             # HITRAN reference temperature [K]
T0 = 273.15              # 0 Celsius [K]
Pmin = 1e-9
