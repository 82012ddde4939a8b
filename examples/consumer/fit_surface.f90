!> A caller of the drop-in module the way an existing SPLPAK user would write it (compare the reference's
!! README.md:44-53 and test/splpak_test.f90): fit a surface to scattered samples, evaluate it and its gradient.
!! `fit_surface host` selects the module's host solver (no GPU needed); without the argument the fit runs on
!! the MI355X.  Built against splpak_amd/fortran/build/libsplpak.so by the Makefile next to this file.
program fit_surface
    use splpak_module, wp => splpak_wp
    implicit none
    integer,parameter :: m = 5000, nod = 12
    integer :: nodes(2), ierror, i
    real(wp) :: xdata(2,m), ydata(m), xmin(2), xmax(2), coef(nod*nod), work(nod*nod*(nod*nod+1)), x(2), f, fx, u(2)
    character(len=16) :: arg
    type(splpak_type) :: s

    call get_command_argument(1, arg)
    if (trim(arg) == 'host') call s%set_host(.true.)
    call random_seed()
    do i = 1, m
        call random_number(u)
        xdata(:,i) = u
        ydata(i) = sin(3.0_wp*u(1))*cos(2.0_wp*u(2))
    end do
    xmin = 0.0_wp; xmax = 1.0_wp; nodes = nod
    call s%initialize(2,xdata,2,ydata,m,xmin,xmax,nodes,1.0_wp,coef,size(coef),work,size(work),ierror)
    if (ierror /= 0) error stop 'fit failed'
    x = [0.4_wp, 0.6_wp]
    f = s%evaluate(2,x,coef,xmin,xmax,nodes,ierror)
    fx = s%evaluate(2,x,[1,0],coef,xmin,xmax,nodes,ierror)
    write(*,'(A,F10.6,A,F10.6)') ' f(0.4,0.6) = ', f, '   exact ', sin(1.2_wp)*cos(1.2_wp)
    write(*,'(A,F10.6,A,F10.6)') ' df/dx1     = ', fx, '   exact ', 3.0_wp*cos(1.2_wp)*cos(1.2_wp)
    if (abs(f - sin(1.2_wp)*cos(1.2_wp)) > 1.0e-3_wp) error stop 'value off'
    write(*,'(A)') ' OK fit_surface'
end program fit_surface
